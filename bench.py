#!/usr/bin/env python3
"""bench.py -- rays/s of the NeRF render hot path (BASELINE.json metric) on N MI355X GPUs of one node.

One "step" = one pass of the hot path over one synthetic frame: 256x256 rays, 64 coarse + 128 fine
samples (ray generation -> stratified z -> coarse PE+MLP -> composite -> inverse-CDF resample ->
fine PE+MLP on 192 sorted samples -> composite -> RGB), inputs resident in HBM (rays are generated on
device; weights uploaded before the timed region).  With N > 1 the frame's rays are sharded into N
contiguous slabs (one process per GPU) and ONE all-gather (RCCL over xGMI) assembles the RGB image on
every rank inside the timed region: strong scaling of the named config.

Prints ONE JSON line on rank 0.  `roofline` prices the dominant kernel (fused PE+MLP, MFMA-bound) from
HIP events recorded on the kernel's own stream inside the timed region; `cpu_baseline` times the CPU
oracle (numpy/OpenBLAS restatement of the reference algorithm -- NOT TensorFlow, which is not
installable here) on a bounded sample of the same workload on the host cores of rank 0.
"""
import argparse
import hashlib
import json
import os
import signal
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FLOPS_PER_ROW = 2 * 512152          # GEMM MACs per sample x 2
PEAK_TFLOPS = {"f32": 157.3, "f16x3": 2500.0, "f16": 2500.0}   # MI355X_MICROARCH.md: dense MFMA peak per dtype fed to MFMA
PEAK_HBM_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E spec peak (6.3 TB/s is what a float4 copy achieves)
H = W = 256
SC, SF = 64, 128
ALGO_BYTES_PER_FRAME = 256 * 256 * (32 + 12) + 2 * 514332 * 4      # rays + rgb + both networks' weights = 7.0 MB
NEAR, FAR, FOV = 2.0 / 3.0, 5.0 / 3.0, 0.6911112


def sphere_matrix(radius, x_rot, y_rot, z_rot):
    """get_sphere_matrix semantics (src/UtilsCV.py:101-121), restated: Rz Ry Rx T."""
    xr, yr, zr = np.deg2rad([x_rot, y_rot, z_rot])
    t = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, radius], [0, 0, 0, 1]], dtype=np.float64)
    rx = np.array([[1, 0, 0, 0], [0, np.cos(xr), -np.sin(xr), 0], [0, np.sin(xr), np.cos(xr), 0], [0, 0, 0, 1]])
    ry = np.array([[np.cos(yr), 0, -np.sin(yr), 0], [0, 1, 0, 0], [np.sin(yr), 0, np.cos(yr), 0], [0, 0, 0, 1]])
    rz = np.array([[np.cos(zr), -np.sin(zr), 0, 0], [np.sin(zr), np.cos(zr), 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    return (rz @ (ry @ (rx @ t))).astype(np.float32)


def cpu_baseline(blob_c, blob_f, c2w, seconds_budget=20.0, gpu_frame=None):
    """Oracle ("port") on the host cores, on a bounded sample of the same frame."""
    from oracle import nerf_oracle as O
    # the GPU box shows every host core but a 1-GPU job owns a 16-core share: pin BLAS to that
    threads = max(1, min(16, len(os.sched_getaffinity(0))))
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=threads)
    except Exception:
        limiter = None
    coarse, fine = O.unpack_blob(blob_c), O.unpack_blob(blob_f)
    dirs = O.get_rays_directions(H, W, FOV, c2w).reshape(-1, 4)
    orig = np.broadcast_to(c2w[:, 3], dirs.shape).astype(np.float32)

    def run(n):
        pick = np.linspace(0, H * W - 1, n).astype(np.int64)
        uc = O.philox_uniform(0, pick.astype(np.uint64), SC, 0)
        uf = O.philox_uniform(0, pick.astype(np.uint64), SF, 1)
        t0 = time.perf_counter()
        out = O.render(coarse, fine, orig[pick], dirs[pick], NEAR, FAR, uc, uf)
        return time.perf_counter() - t0, pick, out[0]

    t_small, _, _ = run(256)                             # calibration (also warms BLAS threads)
    n = int(min(16384, max(512, 256 * seconds_budget / max(t_small, 1e-3))))
    n = (n // 256) * 256
    t, pick, rgb = run(n)
    if limiter is not None:
        limiter.restore_original_limits()
    # the oracle as the CHECKER of the measured path (never the thing measured): the device frame of seed 0 -- the very
    # draws the oracle made above, Philox keyed by (seed, global ray index) -- against the oracle's rays
    parity = None
    if gpu_frame is not None:
        dev = np.asarray(gpu_frame(0)).reshape(-1, 3)[pick]
        err = float(np.abs(dev - rgb).max())
        parity = {"max_abs_rgb_vs_oracle": err, "rays": int(n), "bar": 1e-4, "ok": bool(err <= 1e-4),
                  "note": "the timed mode's frame of seed 0 against the CPU oracle on the same rays, weights and draws"}
    return {"value": n / t, "unit": "rays/s", "cores": int(threads), "kind": "port",
            "sample": f"{n} rays of the same 256x256 frame (64+128 samples), numpy fp32 + OpenBLAS, "
                      f"{t:.1f} s; CPU restatement of the reference algorithm (not TensorFlow)"}, parity


def csrc_sha16():
    """Hash of the kernel sources the running library was built from (profiles/pmc_traffic.json records the hash its
    counters were collected on, so a stale traffic figure can say so)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "nerf_and_dietnerf_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


# ---------------------------------------------------------------------------------------------------------------------
# `python bench.py --gpus N` starts its own ranks.  The driver launches N > 1 through torch.distributed.run (WORLD_SIZE is
# then set and this block is skipped); a harness that calls `python3 bench.py --gpus 8` the way it calls `--gpus 1` gets
# the same one-process-per-GPU job: N fresh children of THIS process -- started before it touches torch or the GPU,
# never a re-exec -- with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rank 0's stdout (the one JSON line) passed
# through, the other ranks' output on stderr, the worst exit code returned, and the whole group killed when one rank
# fails or the wall-clock limit (BENCH_LAUNCH_TIMEOUT seconds, default 1500) passes.
# ---------------------------------------------------------------------------------------------------------------------
def child_argv(argv):
    """The command line of one rank: this script with the very same arguments."""
    return [sys.executable, os.path.abspath(__file__)] + list(argv)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _kill_group(proc, sig):
    try:
        os.killpg(proc.pid, sig)          # every rank is its own session leader (start_new_session): pid == pgid
    except (ProcessLookupError, PermissionError):
        pass


def launch_ranks(cmd, n, timeout_s=1500.0, env=None, out=None, err=None, poll_s=0.1):
    """Start ``n`` copies of ``cmd`` as ranks 0..n-1 of one node and wait for them.  -> worst exit code (124 = timed out)."""
    out = out or sys.stdout
    err = err or sys.stderr
    base = dict(os.environ if env is None else env)
    base.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs, pumps = [], []

    def pump(stream, sink, tag):
        for line in iter(stream.readline, ""):
            if tag is None and not line.lstrip().startswith("{"):
                # rank 0's stdout carries the ONE JSON line; anything else a library prints there (gloo's connection
                # banner) goes to stderr so that stdout stays machine-readable
                err.write(f"[rank 0] {line}")
                err.flush()
                continue
            sink.write(line if tag is None else f"[rank {tag}] {line}")
            sink.flush()
        stream.close()

    try:
        for r in range(n):
            e = dict(base, RANK=str(r), LOCAL_RANK=str(r), GROUP_RANK="0")
            p = subprocess.Popen(cmd, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, bufsize=1,
                                 start_new_session=True)
            procs.append(p)
            for stream, sink, tag in ((p.stdout, out if r == 0 else err, None if r == 0 else r), (p.stderr, err, r)):
                t = threading.Thread(target=pump, args=(stream, sink, tag), daemon=True)
                t.start()
                pumps.append(t)
        t0 = time.monotonic()
        rc = 0
        while True:
            codes = [p.poll() for p in procs]
            bad = [c for c in codes if c not in (None, 0)]
            if bad:
                rc = max(bad, key=abs)
                err.write(f"bench.py: rank {codes.index(bad[0])} exited with {bad[0]}; stopping the other ranks\n")
                break
            if all(c == 0 for c in codes):
                break
            if time.monotonic() - t0 > timeout_s:
                rc = 124
                err.write(f"bench.py: {n}-rank run exceeded {timeout_s:.0f} s; stopping it\n")
                break
            time.sleep(poll_s)
    except BaseException:
        rc = 130
        raise
    finally:
        live = [p for p in procs if p.poll() is None]
        for p in live:
            _kill_group(p, signal.SIGTERM)
        t1 = time.monotonic()
        while any(p.poll() is None for p in live) and time.monotonic() - t1 < 10.0:
            time.sleep(0.05)
        for p in live:
            if p.poll() is None:
                _kill_group(p, signal.SIGKILL)
        for p in procs:
            try:
                p.wait(timeout=10)
            except Exception:      # noqa: BLE001
                pass
        for t in pumps:
            t.join(timeout=5)
    return rc if rc >= 0 else 128 - rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--precision", default="f16x3", choices=["fp32", "f16x3", "f16"],
                    help="arithmetic of the 256-wide contractions: f16x3 = 3-pass split-fp16 MFMA with fp32 accumulate "
                         "(fp32-class accuracy, tests/test_gpu_parity.py); fp32 = exact fp32 MFMA")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the secondary training-step measurement")
    ap.add_argument("--quick", action="store_true",
                    help="headline + fp32 + fp16 modes only: skip the sustained run, the shipped-weights run and configs 4/5")
    ap.add_argument("--c-gather", action="store_true",
                    help="N>1: assemble the frame with the library's own ncclAllGather (nerf_render_image_sharded) "
                         "instead of torch.distributed.all_gather_into_tensor")
    ap.add_argument("--rehearse-world", type=int, default=0,
                    help="1-GPU rehearsal of the per-rank work at N=<k>: render only rank 0's slab of a k-way split "
                         "(no collective); the printed value is that slab's rays/s x k (an estimate, not a result)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: be the launcher (before torch or the GPU is touched in this process)
        raise SystemExit(launch_ranks(child_argv(sys.argv[1:]), args.gpus,
                                      float(os.environ.get("BENCH_LAUNCH_TIMEOUT", "1500"))))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    # BENCH_BACKEND=gloo lets several ranks share one GPU to rehearse the N>1 path on a 1-GPU box
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    dev_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)

    import nerf_and_dietnerf_amd as N

    blob_c, blob_f = N.glorot_blob(0), N.glorot_blob(1)       # random-init weights of the reference architecture
    net_cfg = {"hidden_layer_dim": 256, "last_hidden_layer_dim": 128, "leaky_relu_alpha": 0.05,
               "n_pos_enc_dim_xyz": 5, "n_pos_enc_view_dir": 4, "n_angles_for_model": 2,
               "n_rays_in_batch_train": 4096, "n_rays_in_batch_render": 4096}
    model = N.NeRF(net_cfg, {"n_render_samples_coarse": SC, "n_render_samples_fine": SF}, NEAR, FAR,
                   device=dev_index, precision=args.precision)
    model.set_weights(blob_c, blob_f)
    model.ctx.use_torch_stream()
    c2w = sphere_matrix(1.0, -30.0, 45.0, 0.0)
    total = H * W
    begin, count = N.ray_slab(total, rank, world)
    if args.rehearse_world > 1 and world == 1:
        begin, count = N.ray_slab(total, 0, args.rehearse_world)

    c_gather = args.c_gather and world > 1 and (backend == "nccl" or bool(os.environ.get("NERF_RCCL_LIB")))
    if c_gather:
        model.ctx.comm_init_from_torch()

    def step(seed):
        if c_gather:
            return model.ctx.render_image_sharded(c2w, FOV, H, W, 1 << 18, SC, SF, seed=seed, device_out=True)
        # whole-slab batch (the library's default); results do not depend on the batch size
        rgb = model.render_image(c2w, FOV, H, W, batch_size_input=1 << 18, seed=seed, ray_begin=begin,
                                 ray_count=count, device_out=True, rgb_only=True)[0]
        if world > 1:
            rgb = N.gather_slabs(rgb, total)
        return rgb

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def ev():
        e = torch.cuda.Event(enable_timing=True)
        e.record()                        # torch's current stream == the library's stream (use_torch_stream)
        return e

    def step_marked(seed, marks):
        """One step with stream events around its two halves: render of this rank's slab, then the gather."""
        e0 = ev()
        if c_gather:
            rgb = model.ctx.render_image_sharded(c2w, FOV, H, W, 1 << 18, SC, SF, seed=seed, device_out=True)
            e1 = e2 = ev()                # render + ncclAllGather are one library call: not separable here
        else:
            rgb = model.render_image(c2w, FOV, H, W, batch_size_input=1 << 18, seed=seed, ray_begin=begin,
                                     ray_count=count, device_out=True, rgb_only=True)[0]
            e1 = ev()
            if world > 1:
                rgb = N.gather_slabs(rgb, total)
            e2 = ev()
        marks.append((e0, e1, e2))
        return rgb

    gather_check = None
    c_check_hung = False          # a worker thread is still blocked inside RCCL: leave through os._exit at the end
    c_check_ran = False           # the second communicator was attempted in this process: no torch teardown afterwards
    c_check_rc = 0

    def c_level_check(chk, need_init):
        """The two assemblies of the frame -- torch's collective and the library's own ncclAllGather
        (nerf_render_image_sharded) -- must agree bit for bit.  The library's communicator is a SECOND RCCL communicator
        beside torch's and has never met real peers on this pool (one-GPU boxes; the rehearsals use tests/stub_rccl.c), so
        everything that touches it runs in a worker thread under a wall-clock limit (BENCH_C_CHECK_TIMEOUT seconds, default
        120): a communicator that cannot be created is reported, one that never returns costs the check -- and, since
        round 4, NOTHING ELSE: without --c-gather the check runs AFTER every measurement and after the last torch
        collective this run needs, so even a second communicator that wedges the device cannot take the scaling
        numbers with it.  -> (check string, hung, mismatch)"""
        # torch's assembly first, in the main thread (needs nothing from the library's communicator)
        b_img = N.gather_slabs(model.render_image(c2w, FOV, H, W, batch_size_input=1 << 18, seed=12345, ray_begin=begin,
                                                  ray_count=count, device_out=True, rgb_only=True)[0], total)
        uid = None
        if need_init:
            ids = [chk.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            uid = ids[0]
        torch.cuda.synchronize()
        box = {}

        def c_level_frame():
            try:
                if os.environ.get("BENCH_C_CHECK_TEST_HANG") == "1":     # test hook: a call that never returns (touches nothing)
                    threading.Event().wait()
                torch.cuda.set_device(dev_index)                 # (torch's current device is per thread)
                if uid is not None:
                    chk.comm_init(uid, rank, world)
                a_img = chk.render_image_sharded(c2w, FOV, H, W, 1 << 18, SC, SF, seed=12345, device_out=True)
                chk.synchronize()
                torch.cuda.synchronize()
                box["equal"] = bool(torch.equal(a_img.reshape(-1, 3), b_img.reshape(-1, 3)))
                if not box["equal"]:
                    box["diff"] = float((a_img.reshape(-1, 3) - b_img.reshape(-1, 3)).abs().max())
            except Exception as e:                               # noqa: BLE001
                box["err"] = e

        worker = threading.Thread(target=c_level_frame, daemon=True)
        worker.start()
        worker.join(float(os.environ.get("BENCH_C_CHECK_TIMEOUT", "120")))
        if worker.is_alive():
            return ("c-level communicator / first ncclAllGather did not return within the time limit; torch all-gather "
                    "only"), True, False
        if "err" in box:
            e = box["err"]
            return f"c-level communicator unavailable ({type(e).__name__}: {e}); torch all-gather only", False, False
        if not box["equal"]:
            return (f"MISMATCH: nerf_render_image_sharded (RCCL inside the library) and the torch all-gather assemble "
                    f"different frames on rank {rank} (max diff {box['diff']})"), False, True
        return "c-level ncclAllGather image == torch all_gather image, bit for bit", False, False

    if c_gather:
        # --c-gather times the library's own collective: its communicator exists already (comm_init_from_torch above) and
        # has to be proven before the timed region; a failure here fails the run
        gather_check, c_check_hung, bad = c_level_check(model.ctx, False)
        c_check_ran = True
        if c_check_hung or bad or "unavailable" in gather_check:
            sys.stderr.write(f"rank {rank}: {gather_check}\n")
            sys.stderr.flush()
            os._exit(3)
        gather_check += " (checked in warm-up)"

    for i in range(args.warmup):
        img = step(i)
    sync()
    marks = []
    model.ctx.enable_timing(True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        img = step_marked(args.warmup + i, marks)
    sync()
    elapsed = time.perf_counter() - t0
    mlp_ms, n_launch, n_rows = model.ctx.read_timing()
    model.ctx.enable_timing(False)
    # per-step GPU times from the stream events (no host synchronisation inside the timed region)
    step_ms = [marks[i][0].elapsed_time(marks[i + 1][0]) for i in range(len(marks) - 1)]
    render_ms = [m[0].elapsed_time(m[1]) for m in marks]
    gather_ms = [m[1].elapsed_time(m[2]) for m in marks]
    gap_ms = [marks[i][2].elapsed_time(marks[i + 1][0]) for i in range(len(marks) - 1)]   # stream idle between steps
    med = lambda v: float(np.median(v)) if len(v) else None       # noqa: E731
    attribution = {"rank": rank, "slab_rays": int(count), "render_ms_median": med(render_ms),
                   "mlp_kernel_ms_per_step": mlp_ms / max(args.steps, 1), "gather_ms_median": med(gather_ms),
                   "host_gap_ms_median": med(gap_ms), "step_ms_median": med(step_ms)}
    per_rank = [attribution]
    if world > 1:
        box = [None] * world
        dist.all_gather_object(box, attribution)
        per_rank = box

    # K frames in flight between gathers: separates launch latency at small slabs from gather bandwidth (N>1 only)
    in_flight = None
    if world > 1 and not c_gather:
        K = 8
        def step_k(seed0):
            parts = [model.render_image(c2w, FOV, H, W, batch_size_input=1 << 18, seed=seed0 + k, ray_begin=begin,
                                        ray_count=count, device_out=True, rgb_only=True)[0] for k in range(K)]
            slab = torch.stack(parts, dim=1)                       # (rays of the slab, K, 3): one gather for K frames
            return N.gather_slabs(slab, total)
        step_k(0)
        sync()
        tk = time.perf_counter()
        reps = max(1, args.steps // 2)
        for i in range(reps):
            out_k = step_k(1000 + i * K)
        sync()
        ek = time.perf_counter() - tk
        tt = torch.tensor([ek], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        ek = float(tt.item())
        assert tuple(out_k.shape) == (total, K, 3)
        in_flight = {"frames_per_gather": K, "value": total * K * reps / ek, "unit": "rays/s",
                     "ms_per_frame": ek / (K * reps) * 1e3,
                     "note": "K frames of this rank's slab rendered back to back, ONE all-gather of (rays, K, 3); "
                             "extra figure, not the headline"}
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert bool(torch.isfinite(img).all()) and tuple(img.shape[-1:]) == (3,)
    if args.rehearse_world > 1:
        args.no_cpu_baseline = True
    # (also under the one-GPU rehearsal -- BENCH_BACKEND=gloo with NERF_RCCL_LIB naming tests/stub_rccl.c's stand-in, the only
    # way two ranks can share a device -- so that this check itself has run before the driver's first real N > 1 launch)
    if world > 1 and not c_gather and (backend == "nccl" or os.environ.get("NERF_RCCL_LIB")):
        # every measurement is taken and every torch collective this run needs is behind us: now the library's own
        # communicator may be tried, on a context of ITS OWN (same weights, own stream; a ctx is single-caller)
        chk = N.Context(near=NEAR, far=FAR, n_angles=2, precision=args.precision, device=dev_index)
        chk.load_weights(0, blob_c); chk.load_weights(1, blob_f)
        gather_check, c_check_hung, bad = c_level_check(chk, True)
        c_check_ran = True
        c_check_rc = 1 if bad else 0
        gather_check += " (checked after the timed region)"

    def roof(kernel, rows, ms, dtype_key):
        a = rows * FLOPS_PER_ROW / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        return {"bound": "mfma", "kernel": kernel, "achieved": a, "peak": PEAK_TFLOPS[dtype_key], "unit": "TFLOP/s",
                "frac": a / PEAK_TFLOPS[dtype_key], "rows": int(rows), "flops_per_row": FLOPS_PER_ROW}

    one_tile = os.environ.get("NERF_F16_TILES", "") == "1"
    KERNEL = {"fp32": "mlp_fp32_kernel", "f16x3": "mlp_f16x3_kernel",
              "f16": "mlp_f16x3_kernel<single_pass>" if one_tile else "mlp_f16_2t_kernel"}
    DKEY = {"fp32": "f32", "f16x3": "f16x3", "f16": "f16"}

    def side_run(mdl, precision, k, h, w, sc, sf, pose, fov, batch=1 << 18, min_seconds=0.0):
        """k frames (or at least min_seconds of them) of (h, w, sc+sf) in `precision` on `mdl`: rays/s, median step,
        roofline of the fused kernel from the library's own HIP events."""
        mdl.ctx.set_precision(precision)
        f = lambda sd: mdl.render_image(pose, fov, h, w, batch_size_input=batch, n_render_samples_c=sc,   # noqa: E731
                                        n_render_samples_f=sf, seed=sd, device_out=True, rgb_only=True, honor_batch=True)[0]
        f(0)
        sync()
        mdl.ctx.enable_timing(True)
        evs = [ev()]
        t = time.perf_counter()
        n = 0
        while n < k or (time.perf_counter() - t) < min_seconds:
            f(n + 1)
            evs.append(ev())
            n += 1
            if min_seconds > 0 and n % 16 == 0:
                torch.cuda.synchronize()          # bound the queue while filling a wall-clock budget
        sync()
        e = time.perf_counter() - t
        ms, nl, rows = mdl.ctx.read_timing()
        mdl.ctx.enable_timing(False)
        per = [evs[i].elapsed_time(evs[i + 1]) for i in range(n)]
        rows_per_ray = sc + (sc + sf if sf else 0)
        return {"value": h * w * n / e, "unit": "rays/s", "steps": n, "seconds": e, "ms_per_step": e / n * 1e3,
                "ms_per_step_median": float(np.median(per)), "mlp_rows_per_ray": rows_per_ray,
                "flops_per_ray": rows_per_ray * FLOPS_PER_ROW,
                "roofline": dict(roof(KERNEL[precision], rows, ms, DKEY[precision]), launches=int(nl),
                                 avg_launch_ms=ms / max(nl, 1))}

    other = fp16_mode = sustained = shipped = cfg4 = cfg5 = None
    if world == 1 and args.precision == "f16x3" and args.rehearse_world <= 1:
        # the exact-fp32 mode and the single-pass fp16 mode of the same step, beside the headline mode
        other = dict(side_run(model, "fp32", 3, H, W, SC, SF, c2w, FOV), dtype="f32")
        other["roofline"]["frac_vs_fp32_matrix_peak"] = other["roofline"]["frac"]
        fp16_mode = dict(side_run(model, "f16", 5, H, W, SC, SF, c2w, FOV),
                         dtype="f16 (single pass, fp16 activations between layers: mixed_float16-class, not the fp32 parity mode)")
        if not args.quick:
            # >= 10 s of back-to-back frames in the headline mode: the sustained (thermal / DVFS steady-state) rate
            sustained = dict(side_run(model, "f16x3", 1, H, W, SC, SF, c2w, FOV, min_seconds=10.0),
                             note="headline mode, same frame, >= 10 s of wall clock")
            # SURVEY 8d input set (A): the reference's shipped epoch-95 weights (operand statistics of a trained net)
            ck = os.path.join(ROOT, "tests", "golden", "alexander50_epoch095.npz")
            if os.path.exists(ck):
                g = np.load(ck)
                m_a = N.NeRF(net_cfg, {"n_render_samples_coarse": SC, "n_render_samples_fine": SF}, float(g["near"]),
                             float(g["far"]), device=dev_index, precision="f16x3")
                m_a.set_weights(g["blob_coarse"], g["blob_fine"])
                m_a.ctx.use_torch_stream()
                shipped = dict(side_run(m_a, "f16x3", args.steps, H, W, SC, SF, g["c2w_test"], float(g["fov"])),
                               weights="reference's shipped Alexander epoch-95 checkpoint (data fixture), its test-view "
                                       "pose / bounds / fov at 256x256; headline mode")
                shipped["fp16_single_pass"] = side_run(m_a, "f16", 5, H, W, SC, SF, g["c2w_test"], float(g["fov"]))
                m_a.ctx.close()
            # BASELINE configs[3]: DietNeRF consistency-render shape (150x150, 55+55, batches of 2048 rays)
            cfg4 = dict(side_run(model, "f16x3", 20, 150, 150, 55, 55, c2w, FOV, batch=2048),
                        workload="150x150, 55 coarse + 55 fine, 2048-ray batches (src/DietNeRF.py:215-218), headline mode")
            # ... and what DietNeRF does with that image: back-propagate dL/d(image) through NeRF.render, batch by batch --
            # under both policies (the reference can only run it under mixed_float16, src/ExecutionRun.py:220-221)
            model.ctx.set_precision("f16x3")
            d_img = torch.rand((150 * 150, 3), device="cuda") * 1e-3
            dirs4 = model.ctx.get_rays_directions(150, 150, FOV, torch.as_tensor(c2w, device="cuda")).reshape(-1, 4)
            orig4 = torch.as_tensor(c2w[:, 3], device="cuda").expand(150 * 150, 4).contiguous()

            def consistency_backward(seed):
                for b in range(0, 150 * 150, 2048):
                    model.ctx.train_render_gradients(orig4[b:b + 2048], dirs4[b:b + 2048], d_img[b:b + 2048], 55, 55,
                                                     seed=seed, ray_base=b, accumulate=b > 0)

            def time_consistency(mixed):
                model.compile(5e-4, mixed_float16=mixed)
                consistency_backward(0)
                sync()
                t4 = time.perf_counter()
                for i in range(3):
                    consistency_backward(1 + i)
                    model.ctx.train_apply()
                sync()
                e = (time.perf_counter() - t4) / 3
                ls = model.ctx.train_loss_scale()
                model.ctx.train_end()
                return {"ms_per_image": e * 1e3, "value": 150 * 150 / e, "unit": "rays/s", "steps_applied": ls[1],
                        "steps_skipped": ls[2]}
            rg32, rg16 = time_consistency(False), time_consistency(True)
            cfg4["render_gradients"] = dict(rg32, policy="float32",
                                            note="nerf_train_render_gradients: forward with stash + backward through the "
                                                 "merged 110-sample fine pass and the sampler, 11 batches of 2048 rays "
                                                 "(src/DietNeRF.py:204-222), then nerf_train_apply",
                                            mixed_float16_policy=dict(rg16, note="the reference's production policy: d_rgb "
                                                                      "loss-scaled on the device, single-pass fp16 chain, "
                                                                      "unscaled gradients, one verdict per image"))
            # ... and the whole DietNeRF.train_step cycle through the model class (nerf_and_dietnerf_amd/dietnerf.py): 2048-ray
            # batches x (64 + 128) (256px_alexander_71pics_sphere_dietnerf.yaml), every 13th step with the 150x150 x (55 + 55)
            # consistency render + backward.  The embedding network is the caller's (the reference's ViT-B/32 is a TF-Hub
            # fetch): a small conv stand-in here, so the cycle time is the LIBRARY's share of a DietNeRF cycle
            def dietnerf_cycle(mixed):
                conv = torch.nn.Conv2d(3, 16, 16, 16).cuda()
                lin = torch.nn.Linear(16 * 14 * 14, 64).cuda()
                for prm in list(conv.parameters()) + list(lin.parameters()):
                    prm.requires_grad_(False)
                emb = lambda x: lin(torch.tanh(conv(x.permute(0, 3, 1, 2))).flatten(1))       # noqa: E731
                gen4 = torch.Generator(device="cuda").manual_seed(4)
                imgs = torch.rand((4, 64, 64, 3), device="cuda", generator=gen4)
                poses = np.stack([sphere_matrix(1.0, -30.0 - 10 * i, 45.0 + 20 * i, 0.0) for i in range(4)])
                dn = N.DietNeRF(dict(net_cfg, n_rays_in_batch_train=2048), {"n_render_samples_coarse": SC,
                                "n_render_samples_fine": SF}, NEAR, FAR, imgs, poses, FOV, embedder=emb, device=dev_index)
                dn.set_weights(blob_c, blob_f)
                dn.compile(5e-4, mixed_float16=mixed)
                b_o = torch.zeros((2048, 4), device="cuda"); b_o[:, 2] = 1.0; b_o[:, 3] = 1.0
                b_d = torch.randn((2048, 4), device="cuda", generator=gen4) * 0.3; b_d[:, 2] = -1.0; b_d[:, 3] = 0.0
                b_t = torch.rand((2048, 3), device="cuda", generator=gen4)
                for _ in range(13):                           # one warm-up cycle (its 13th step is a consistency step)
                    dn.train_step((b_o, b_d, b_t), want_metrics=False)
                sync()
                t_plain = time.perf_counter()
                for _ in range(12):
                    dn.train_step((b_o, b_d, b_t), want_metrics=False)
                sync()
                t_cs = time.perf_counter()
                dn.train_step((b_o, b_d, b_t), want_metrics=False)       # step 26: ray loss + consistency loss
                sync()
                t_end = time.perf_counter()
                ls = dn.ctx.train_loss_scale()
                dn.ctx.train_end(); dn.ctx.close()
                plain, cs = (t_cs - t_plain) / 12 * 1e3, (t_end - t_cs) * 1e3
                return {"ms_per_plain_step": plain, "ms_per_consistency_step": cs, "ms_per_13_step_cycle": 12 * plain + cs,
                        "rays_per_cycle": 13 * 2048 + 150 * 150, "steps_applied": ls[1], "steps_skipped": ls[2]}
            cfg4["dietnerf_train_cycle"] = {
                "float32": dietnerf_cycle(False), "mixed_float16": dietnerf_cycle(True),
                "note": "DietNeRF.train_step through nerf_and_dietnerf_amd.DietNeRF: 2048-ray batches x (64 + 128), every 13th "
                        "step + a 150x150 x (55 + 55) source render and its backward through NeRF.render in 2048-ray batches "
                        "(src/DietNeRF.py:120-222); embedder = a small conv stand-in (the reference's ViT-B/32 is a remote "
                        "fetch), so these are the library's share of the cycle"}
            # BASELINE configs[4]: 800x800, 64 coarse + 256 fine, fp16 MLP
            cfg5 = dict(side_run(model, "f16", 3, 800, 800, 64, 256, c2w, FOV),
                        workload="800x800, 64 coarse + 256 fine (fine pass 320 samples), single-pass fp16 MLP mode")
            cfg5["f16x3_mode"] = side_run(model, "f16x3", 2, 800, 800, 64, 256, c2w, FOV)
        model.ctx.set_precision("f16x3")

    # the training step on the reference's batch, timed briefly beside the headline (N=1 only)
    train = None
    if world == 1 and args.rehearse_world <= 1 and not args.no_train:
        n_tr, k_tr = 4096, 30
        gen = torch.Generator(device="cuda").manual_seed(0)
        t_o = torch.zeros((n_tr, 4), device="cuda"); t_o[:, 2] = 1.0; t_o[:, 3] = 1.0
        t_d = torch.randn((n_tr, 4), device="cuda", generator=gen) * 0.3; t_d[:, 2] = -1.0; t_d[:, 3] = 0.0
        t_rgb = torch.rand((n_tr, 3), device="cuda", generator=gen)
        model.compile(5e-4)
        for i in range(3):
            model.ctx.train_step(t_o, t_d, t_rgb, SC, SF, seed=i, want_metrics=False)
        sync()
        t2 = time.perf_counter()
        for i in range(k_tr):
            model.ctx.train_step(t_o, t_d, t_rgb, SC, SF, seed=10 + i, want_metrics=False)
        sync()
        e_tr = (time.perf_counter() - t2) / k_tr
        m_tr = model.ctx.train_step(t_o, t_d, t_rgb, SC, SF, seed=99)
        model.ctx.train_end()
        # the reference's production policy (mixed_float16 + dynamic loss scaling) on the same batch
        model.compile(5e-4, mixed_float16=True)
        for i in range(3):
            model.ctx.train_step(t_o, t_d, t_rgb, SC, SF, seed=i, want_metrics=False)
        sync()
        t2m = time.perf_counter()
        for i in range(k_tr):
            model.ctx.train_step(t_o, t_d, t_rgb, SC, SF, seed=10 + i, want_metrics=False)
        sync()
        e_mx = (time.perf_counter() - t2m) / k_tr
        ls_mx = model.ctx.train_loss_scale()
        model.ctx.train_end()
        tf_tr = 3 * n_tr * (SC + SF) * FLOPS_PER_ROW / e_tr / 1e12
        tf_mx = 3 * n_tr * (SC + SF) * FLOPS_PER_ROW / e_mx / 1e12
        # HBM bytes of one step: rocprofv3 PMC passes collected OFFLINE over tools/train_bench.py (same batch), summed over
        # the step's kernels by tools/pmc_summary_r3.py into profiles/pmc_traffic.json
        tr_bytes = {}
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                tr_bytes = json.load(f)
        except Exception:
            pass
        # what a step must move at the least with this three-kernel structure (stash forward, fused data-gradient chain,
        # weight-gradient GEMMs): every 256-wide activation and pre-activation gradient is written once and read once
        rows_tr = n_tr * (SC + SF)
        elems_per_row = 7 * 256 + 320 + 288 + 128 + 8 * 256 + 128      # stash (incl. the two concat buffers) + D buffers

        def hbm_view(seconds, key, elem_bytes):
            by = tr_bytes.get(key)
            algo = rows_tr * elems_per_row * elem_bytes * 2.0
            measured_on = tr_bytes.get("csrc_sha16")
            return {"bound": "hbm", "bound_status": "priced against HBM, not proven to be bound by it: see bound_evidence",
                    "achieved": (by / seconds / 1e9) if by else None, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": (by / seconds / 1e9 / PEAK_HBM_GBS) if by else None, "traffic": by,
                    "traffic_source": "offline rocprofv3 PMC, summed over the kernels of one step (profiles/pmc_traffic.json)",
                    # the counters were collected on a particular build: say so, and say when the running sources differ
                    "traffic_measured_on_csrc_sha16": measured_on, "running_csrc_sha16": csrc_sha16(),
                    "traffic_stale": (measured_on != csrc_sha16()) if measured_on else None,
                    "bound_evidence": tr_bytes.get("issue_side"),
                    "structural_bytes": algo, "traffic_vs_structural": (by / algo) if by else None,
                    # half of the step's bytes are WRITES (stash, gradient buffers), and the write path saturates well below
                    # the 8 TB/s spec: a bare MFMA chain streaming stores out measures 5.0-5.6 TB/s chip-wide
                    # (tools/microbench/mem_issue_cost, profiles/r3_mem_issue_cost.txt); a float4 copy reads at 6.3 TB/s
                    "achievable": {"write_GBps": 5300.0, "read_GBps": 6300.0,
                                   "frac_of_mean": (by / seconds / 1e9 / 5800.0) if by else None,
                                   "source": "tools/microbench/mem_issue_cost (writes), MI355X_MICROARCH.md (reads)"},
                    "note": "the step's traffic: activations and pre-activation gradients are written once (stash forward, "
                            "backward chain) and read once (weight-gradient GEMMs); structural_bytes counts exactly that. "
                            "The weight-gradient GEMMs stream at 0.8-0.94 of the achievable read rate; the fused stash / "
                            "backward kernels write at 0.7-0.8 of the achievable write rate while their waves are also "
                            "VALU-issue limited (bound_evidence): a mixed bound. SURVEY 8(d)-style algorithmic bytes "
                            "(rays + weights + moments) are ~20 MB."}
        train = {"metric": "train_step (NeRF.train_step: coarse+fine forward, backward incl. sampler, Adam)",
                 "value": n_tr / e_tr, "unit": "rays/s", "ms_per_step": e_tr * 1e3, "steps": k_tr, "dtype": "f16 (3-pass hi/lo split operands, f32 accumulate; fp32-class results)",
                 "batch_rays": n_tr, "samples": f"{SC} coarse + {SF} fine (fine pass on the new samples only)",
                 "loss_finite": bool(m_tr["loss"] == m_tr["loss"]),
                 "policy": "float32 (fp32-class gradients; the figures of this level)",
                 "ms_per_step_reference_policy": e_mx * 1e3,     # = mixed_float16_policy.ms_per_step, the policy the reference trains under
                 "mixed_float16_policy": {"ms_per_step": e_mx * 1e3, "value": n_tr / e_mx, "unit": "rays/s",
                                          "loss_scale": ls_mx[0], "steps_skipped": ls_mx[2],
                                          "note": "the reference's production policy (src/ExecutionRun.py:220-221, "
                                                  "src/NeRF.py:159-163): single-pass fp16 forward, data and weight "
                                                  "gradients on fp16 activation / gradient buffers, fp32 accumulation "
                                                  "and master weights, dynamic loss scaling",
                                          "roofline": dict(hbm_view(e_mx, "train_step_mixed", 2),
                                                           mfma_view={"achieved": tf_mx, "peak": PEAK_TFLOPS["f16"], "unit": "TFLOP/s",
                                                                      "frac": tf_mx / PEAK_TFLOPS["f16"], "mfma_passes_per_product": 1,
                                                                      "flops": "3 x forward GEMM flops"})},
                 "roofline": dict(hbm_view(e_tr, "train_step_f32", 4),
                                  kernel="whole step: mlp_f16x3_stash_kernel (fused forward) + mlp_bwd_f16x3[_dx]_kernel "
                                         "(fused data-gradient chain, pair16 gradient buffers) + gemm_atb_p (weight gradients, batched per pass; the sigma head rides in layer 8's), "
                                         "all 3-pass split-fp16 MFMA with fp32 accumulation",
                                  mfma_view={"achieved": tf_tr, "peak": PEAK_TFLOPS["f16x3"], "unit": "TFLOP/s",
                                             "frac": tf_tr / PEAK_TFLOPS["f16x3"], "frac_vs_fp32_matrix_peak": tf_tr / PEAK_TFLOPS["f32"],
                                             "mfma_passes_per_product": 3,
                                             "flops": "3 x forward GEMM flops (forward, data gradient, weight gradient)"})}

    # the drop-in boundary itself: the host-memory entry point (numpy in, numpy out), one synchronous call per frame --
    # page-locked outputs, copies on a second stream; next to the same call with device-resident outputs (N=1 only)
    host_boundary = None
    if world == 1 and args.rehearse_world <= 1 and not args.quick:
        def frames_per_s(f, k):
            f(100), f(101)
            t_ = time.perf_counter()
            for i in range(k):
                f(i)
            return k / (time.perf_counter() - t_)

        def f_dev(i):
            model.render_image(c2w, FOV, H, W, seed=i, rgb_only=True, device_out=True)
            sync()
        model.ctx.set_precision(args.precision)
        r_dev = frames_per_s(f_dev, 8) * total
        r_rgb = frames_per_s(lambda i: model.render_image(c2w, FOV, H, W, seed=i, rgb_only=True), 8) * total
        r_six = frames_per_s(lambda i: model.render_image(c2w, FOV, H, W, seed=i), 5) * total
        host_boundary = {"unit": "rays/s", "calls": "synchronous, one per 256x256 frame (the steady-state headline enqueues frames back to back)",
                         "device_resident_rgb": r_dev, "host_rgb_only": r_rgb, "host_six_outputs": r_six,
                         "host_rgb_only_vs_device": r_rgb / r_dev, "host_six_outputs_vs_device": r_six / r_dev,
                         "bytes_to_host_per_frame": {"rgb_only": total * 12, "six_outputs": total * (12 + (SC + SF) * 28)},
                         "note": "nerf_render_image(NERF_MEM_HOST) into page-locked buffers (nerf_host_alloc): outputs leave on a "
                                 "copy stream behind per-batch events while the next batch computes; never the headline value"}

    # the xyz-only network (n_angles_for_model = 0, src/NeRF.py:248-288; 5 of the 46 shipped configs) beside the
    # view-direction network: render rate and training step -- reported figures (the -m gpu suite only keeps sanity floors)
    xyz_only = None
    if world == 1 and args.rehearse_world <= 1 and not args.quick:
        xyz_only = {}
        for na in (2, 0):
            cx = N.Context(near=NEAR, far=FAR, n_angles=na, precision="f16x3", device=dev_index)
            cx.load_weights(0, N.glorot_blob(11, n_angles=na))
            cx.load_weights(1, N.glorot_blob(12, n_angles=na))
            cx.use_torch_stream()
            f = lambda sd: cx.render_image(c2w, FOV, H, W, 0, SC, SF, seed=sd, device_out=True, rgb_only=True)   # noqa: E731
            f(0)
            sync()
            tx = time.perf_counter()
            for i in range(6):
                f(1 + i)
            sync()
            r_render = 6 * total / (time.perf_counter() - tx)
            steps_ms = {}
            if not args.no_train:
                gen = torch.Generator(device="cuda").manual_seed(1)
                x_o = torch.zeros((4096, 4), device="cuda"); x_o[:, 2] = 1.0; x_o[:, 3] = 1.0
                x_d = torch.randn((4096, 4), device="cuda", generator=gen) * 0.3; x_d[:, 2] = -1.0; x_d[:, 3] = 0.0
                x_t = torch.rand((4096, 3), device="cuda", generator=gen)
                for pol in ("float32", "mixed_float16"):
                    cx.train_begin(5e-4, mixed_float16=pol == "mixed_float16")
                    for i in range(3):
                        cx.train_step(x_o, x_d, x_t, SC, SF, seed=i, want_metrics=False)
                    sync()
                    tx = time.perf_counter()
                    for i in range(15):
                        cx.train_step(x_o, x_d, x_t, SC, SF, seed=10 + i, want_metrics=False)
                    sync()
                    steps_ms[pol] = (time.perf_counter() - tx) / 15 * 1e3
                    cx.train_end()
            xyz_only["view_direction_network" if na else "xyz_only_network"] = {
                "render_rays_per_s": r_render, "train_ms_per_step": steps_ms or None}
            cx.close()
        a, b = xyz_only["xyz_only_network"], xyz_only["view_direction_network"]
        xyz_only["render_rate_vs_view_direction_network"] = a["render_rays_per_s"] / b["render_rays_per_s"]
        if a["train_ms_per_step"]:
            xyz_only["train_step_time_vs_view_direction_network"] = {
                k: a["train_ms_per_step"][k] / b["train_ms_per_step"][k] for k in a["train_ms_per_step"]}
        xyz_only["note"] = ("256x256, 64 + 128, f16x3 render mode; 4096-ray training steps; the xyz-only network has 12 Dense "
                            "layers, 6 % more MACs per row and one more layer of stash traffic")

    if rank == 0:
        value = total * args.steps / elapsed
        if args.rehearse_world > 1 and world == 1:
            value = count * args.rehearse_world * args.steps / elapsed
        dtype = {"fp32": "f32", "f16x3": "f16x3", "f16": "f16"}[args.precision]
        ach = (n_rows * FLOPS_PER_ROW) / (mlp_ms * 1e-3) / 1e12 if mlp_ms > 0 else 0.0
        # roofline.traffic: HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; gfx950 correction),
        # collected OFFLINE for the N=1 launches and kept in profiles/pmc_traffic.json -- PMC counters cannot be read
        # from inside this process; smaller slabs at N>1: not measured
        traffic = None
        try:
            if world > 1:
                raise LookupError
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                traffic = json.load(f).get(dtype)
        except Exception:
            pass
        out = {
            "metric": "rays/sec (coarse+fine) at 256x256, 64 coarse + 128 fine samples",
            "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "ms_per_step_median": med(step_ms),
            "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None,
            "dtype": {"f32": "f32", "f16x3": "f16 (3-pass hi/lo split operands, f32 accumulate; fp32-class results)",
                      "f16": "f16 (single pass, f32 accumulate, fp16 activations: mixed_float16-class results, NOT the fp32 parity mode)"}[dtype],
            "data": "synthetic (random-init Glorot weights, sphere pose, "
                                                          "on-device Philox draws)",
            "config": {"workload": "256x256 synthetic scene, 64 coarse + 128 fine (BASELINE configs[1]); contractions in " + {"f32": "exact fp32 MFMA", "f16x3": "3-pass split-fp16 MFMA, fp32 accumulate", "f16": "1-pass fp16 MFMA, fp32 accumulate"}[dtype],
                       "rays_per_step": total, "mlp_rows_per_ray": SC + SC + SF,
                       "parallelism": f"ray-sharded x{world}, one all-gather of RGB per frame"},
            "roofline": {"bound": "mfma", "kernel": {"f32": "mlp_fp32_kernel", "f16x3": "mlp_f16x3_kernel", "f16": "mlp_f16x3_kernel<single_pass>" if os.environ.get("NERF_F16_TILES", "") == "1" else "mlp_f16_2t_kernel"}[dtype] + " (fused PE + 11-layer MLP)",
                         "achieved": ach, "peak": PEAK_TFLOPS[dtype], "unit": "TFLOP/s",
                         "frac": ach / PEAK_TFLOPS[dtype], "frac_vs_fp32_matrix_peak": ach / PEAK_TFLOPS["f32"],
                         "mfma_passes_per_product": 3 if dtype == "f16x3" else 1, "traffic": traffic,
                         "traffic_source": "offline rocprofv3 PMC (profiles/pmc_traffic.json), bytes per average launch",
                         # 32 B/ray in (generated on device: 0 from the host) + 12 B/ray rgb out + one
                         # read of both networks' weights per frame; two launches per frame
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_FRAME / 2,
                         "traffic_vs_algorithmic": (traffic / (ALGO_BYTES_PER_FRAME / 2)) if traffic else None,
                         "launches": int(n_launch), "avg_launch_ms": mlp_ms / max(n_launch, 1),
                         "flops_per_row": FLOPS_PER_ROW, "rows": int(n_rows)},
        }
        for key, val in (("fp32_exact_mode", other), ("fp16_single_pass_mode", fp16_mode), ("sustained", sustained),
                         ("shipped_checkpoint_weights", shipped), ("config4_dietnerf_shape", cfg4),
                         ("config5_800x800_fp16", cfg5), ("frames_in_flight", in_flight)):
            if val is not None:
                out[key] = val
        if world > 1:
            out["per_rank"] = per_rank
            out["gather"] = {"kind": "library ncclAllGather (--c-gather)" if c_gather else "torch all_gather_into_tensor",
                             "bytes_per_rank": int(-(-total // world) * 12), "check": gather_check}
        else:
            out["per_rank"] = per_rank
        if train is not None:
            out["training"] = train
        if host_boundary is not None:
            out["host_boundary"] = host_boundary
        if xyz_only is not None:
            out["xyz_only"] = xyz_only
        if world == 1 and not args.no_cpu_baseline:
            model.set_weights(blob_c, blob_f)               # (the training measurement above moved them)
            model.ctx.set_precision(args.precision)
            frame0 = lambda sd: model.render_image(c2w, FOV, H, W, seed=sd, rgb_only=True)[0]     # noqa: E731
            out["cpu_baseline"], parity = cpu_baseline(blob_c, blob_f, c2w, gpu_frame=frame0)
            if parity is not None:
                out["parity_check"] = parity
        print(json.dumps(out), flush=True)
    if world > 1:
        if c_check_ran and not c_gather:
            # a second communicator was tried in this process (and may have failed or hung on this or another rank): no
            # barrier, no RCCL teardown -- every rank leaves by itself; the line is out
            if c_check_rc:
                sys.stderr.write(f"rank {rank}: {gather_check}\n")
            sys.stdout.flush(); sys.stderr.flush()
            os._exit(c_check_rc)
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
