"""Host logic and the C-ABI surface (no GPU compute): the library loads, exports every symbol
include/nerf_mi355.h declares, and fails loudly where it must."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu():
    import torch
    return torch.cuda.is_available()


def test_header_symbols_exported():
    import nerf_and_dietnerf_amd as N
    lib = N._lib.load()
    text = open(os.path.join(ROOT, "include", "nerf_mi355.h")).read()
    declared = set(re.findall(r"\b(nerf_[a-z_]+)\s*\(", text))
    declared -= {"nerf_config", "nerf_outputs", "nerf_ctx"}
    assert len(declared) >= 20
    bound = {name for name, _, _ in N._lib.SYMBOLS}
    assert declared == bound, (declared ^ bound)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.nerf_abi_version() == N._lib.NERF_ABI_VERSION


def test_blob_size_and_config_errors():
    import nerf_and_dietnerf_amd as N
    lib = N._lib.load()
    cfg = N._lib.NerfConfig(5, 4, 2, 256, 128, 0.05, 2.0, 6.0, 0, 0)
    assert lib.nerf_blob_size(ctypes.byref(cfg)) == 514332 == N.blob_size()
    bad = N._lib.NerfConfig(5, 4, 3, 256, 128, 0.05, 2.0, 6.0, 0, 0)
    assert lib.nerf_blob_size(ctypes.byref(bad)) == 0
    assert "should be 1 or 2" in N._lib.last_error()                 # message of src/UtilsCV.py:138
    h = ctypes.c_void_p()
    assert lib.nerf_ctx_create(ctypes.byref(bad), ctypes.byref(h)) != 0 and not h.value


@pytest.mark.skipif(_has_gpu(), reason="checks the no-GPU failure mode")
def test_no_cpu_fallback():
    import nerf_and_dietnerf_amd as N
    with pytest.raises(RuntimeError, match="no CPU path"):
        N.Context()


def test_split_to_batches_matches_reference_semantics(oracle):
    import nerf_and_dietnerf_amd as N
    # src/UtilsNeuralRadianceField.py:32-49: [B]*k (+[-1] remainder), or [total] when total < B
    assert N.get_size_of_splits(4096, 2500) == [2500]
    assert N.get_size_of_splits(4096, 65536) == [4096] * 16
    assert N.get_size_of_splits(2048, 22500) == [2048] * 10 + [-1]
    for b, t in [(4096, 2500), (4096, 65536), (2048, 22500), (7, 7), (3, 10)]:
        assert N.get_size_of_splits(b, t) == oracle.get_size_of_splits(b, t)
        x = np.arange(t * 2, dtype=np.float32).reshape(t, 2)
        parts = N.split_to_batches(x, b)
        assert sum(p.shape[0] for p in parts) == t
        np.testing.assert_array_equal(np.concatenate(parts), x)
        assert [p.shape[0] for p in parts] == [p.shape[0] for p in oracle.split_to_batches(x, b)]
    with pytest.raises(AssertionError):
        N.split_to_batches(np.zeros((4, 2), np.float32), 0)


def test_weights_helpers(oracle):
    import nerf_and_dietnerf_amd as N
    assert N.layer_shapes() == oracle.layer_shapes()
    b = N.glorot_blob(3)
    np.testing.assert_array_equal(b, oracle.glorot_blob(3))
    assert b.size == 514332 and b.dtype == np.float32
    layers = oracle.unpack_blob(b)
    assert all(np.all(bias == 0) for _, bias in layers)
    with pytest.raises(Exception, match="should be 1 or 2"):
        N.layer_shapes(n_angles=3)
    # the xyz-only network (get_network_only_xyz, src/NeRF.py:248-288): 12 Dense layers
    assert N.layer_shapes(n_angles=0) == oracle.layer_shapes(n_angles=0) and len(N.layer_shapes(n_angles=0)) == 12
    assert N.blob_size(n_angles=0) == 577028


def test_ray_slab_partition():
    import nerf_and_dietnerf_amd as N
    for total, world in [(65536, 8), (65536, 1), (2500, 8), (22500, 4), (7, 8), (640000, 8)]:
        slabs = [N.ray_slab(total, r, world) for r in range(world)]
        assert slabs[0][0] == 0
        assert sum(c for _, c in slabs) == total
        for (b0, c0), (b1, _) in zip(slabs, slabs[1:]):
            assert b1 == b0 + c0 or (c0 == 0 and b1 == total)
        per = -(-total // world)
        assert all(c <= per for _, c in slabs)
    assert N.ray_slab(65536, 3, 8) == (3 * 8192, 8192)            # whole rows when H % P == 0


def test_m0_invariant_of_the_fused_kernels(tmp_path):
    """The fused MLP kernels own M0 without a clobber (mlp_common.h::dma_piece*); the Makefile keeps their ISA and
    fails the build when compiler-generated code touches M0.  Here: the checker flags a planted hit, and the ISA
    of the current build (when this tree was built with `make`) is clean."""
    import glob
    import subprocess
    import sys as _sys
    tool = os.path.join(ROOT, "tools", "check_m0.py")
    bad = tmp_path / "bad.s"
    bad.write_text(";;#ASMSTART\n\ts_mov_b32 m0, s4\n;;#ASMEND\n\tv_readlane_b32 s5, v1, m0\n")
    good = tmp_path / "good.s"
    good.write_text(";;#ASMSTART\n\ts_mov_b32 m0, s4\n;;#ASMEND\n\tv_add_f32 v0, v1, v2 ; m0 in a comment\n")
    assert subprocess.run([_sys.executable, tool, str(bad)], capture_output=True).returncode == 1
    assert subprocess.run([_sys.executable, tool, str(good)], capture_output=True).returncode == 0
    isa = sorted(glob.glob(os.path.join(ROOT, "build", "csrc", "mlp_*-hip-amdgcn-amd-amdhsa-gfx950.s")))
    if isa:
        r = subprocess.run([_sys.executable, tool] + isa, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout
        for p in isa:                       # the asm is really in there: the check is not vacuous
            assert "global_load_lds_dwordx4" in open(p).read()


def test_exec_restore_lint(tmp_path):
    """tools/check_exec_restore.py (run by the Makefile on the fused kernels' ISA): flags a vector instruction between the
    join label of an `s_cbranch_execz` and the `s_or_b64 exec` that re-opens the outer mask -- the shape of the hipcc defect
    DESIGN.md section 4.1c describes -- and passes the ISA of the current build."""
    import glob
    import subprocess
    import sys as _sys
    tool = os.path.join(ROOT, "tools", "check_exec_restore.py")
    bad = tmp_path / "bad.s"
    bad.write_text("\ts_and_saveexec_b64 s[0:1], s[40:41]\n\ts_cbranch_execz .LBB0_35\n\tv_mov_b32_e32 v42, v15\n"
                   "\ts_cbranch_execz .LBB0_34\n.LBB0_34:\n\tv_mov_b32_e32 v15, v42\n.LBB0_35:\n"
                   "\ts_or_b64 exec, exec, s[0:1]\n")
    good = tmp_path / "good.s"
    good.write_text("\ts_and_saveexec_b64 s[0:1], s[40:41]\n\ts_cbranch_execz .LBB0_35\n\tv_mov_b32_e32 v42, v15\n"
                    ".LBB0_35:\n\ts_or_b64 exec, exec, s[0:1]\n\tv_mov_b32_e32 v15, v42\n")
    assert subprocess.run([_sys.executable, tool, str(bad)], capture_output=True).returncode == 1
    assert subprocess.run([_sys.executable, tool, str(good)], capture_output=True).returncode == 0
    isa = sorted(glob.glob(os.path.join(ROOT, "build", "csrc", "mlp_*-hip-amdgcn-amd-amdhsa-gfx950.s")))
    if isa:
        r = subprocess.run([_sys.executable, tool] + isa, capture_output=True, text=True)
        assert r.returncode == 0, r.stdout


def test_fragment_major_layout(tmp_path):
    """csrc/frag_layout.h::frag_index -- the element order of the fused trainer's activation / gradient buffers -- is a
    bijection of every 32-row block onto its row-major footprint, puts the 64 (half, sample) lane slots of an 8-feature
    group on consecutive 4-element slots (what makes the fused kernels' stores contiguous), and turns a column offset
    c (c % 8 == 0) into the pointer offset 32 c (what the trainer uses for the encoding columns of the concat buffers)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "frag_check.cpp"
    src.write_text(r"""
#include <cstdio>
#include <vector>
#include "frag_layout.h"
int main() {
    const int lds[] = {64, 128, 256, 288, 320};
    for (int ld : lds) {
        const long long rows = 96;
        std::vector<int> hit(rows * ld, 0);
        for (long long r = 0; r < rows; ++r)
            for (int c = 0; c < ld; ++c) {
                const long long e = nerf::frag_index(r, c, ld);
                if (e < 0 || e >= rows * ld || hit[e]++) { std::printf("not a bijection ld %d row %lld col %d\n", ld, r, c); return 1; }
                if (e / (32 * ld) != r / 32) { std::printf("left its 32-row block\n"); return 1; }
                const long long want = (r / 32) * 32 * ld + (((c / 8) * 64 + ((c / 4) % 2) * 32 + (r % 32)) * 4 + c % 4);
                if (e != want) { std::printf("lane order\n"); return 1; }
                if (c % 8 == 0 && nerf::frag_index(r, c, ld) != nerf::frag_index(r, 0, ld) + 32LL * c) { std::printf("column offset\n"); return 1; }
            }
    }
    std::printf("ok\n");
    return 0;
}
""")
    exe = tmp_path / "frag_check"
    subprocess.run(["g++", "-O1", "-std=c++17", "-I", os.path.join(root, "nerf_and_dietnerf_amd", "csrc"), str(src), "-o", str(exe)],
                   check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stdout + out.stderr


def test_no_wrong_results_switches_in_the_shipped_library():
    """The release library reads the environment only for result-preserving selectors: which kernel family computes the
    same numbers (NERF_TRAIN_*, NERF_F16_TILES) and which librccl carries the bytes (NERF_RCCL_LIB).  Timing
    diagnostics that change results (NERF_DIAG*, NERF_STAMPS, NERF_DIAG_STASH_WRAP) are compile-time flags."""
    import glob
    allowed = re.compile(r"^NERF_(TRAIN_[A-Z_]+|F16_TILES|RCCL_LIB)$")
    seen = set()
    for path in sorted(glob.glob(os.path.join(ROOT, "nerf_and_dietnerf_amd", "csrc", "*"))):
        if not os.path.isfile(path):
            continue
        for m in re.finditer(r"getenv\s*\(\s*([^)]*)\)", open(path, errors="replace").read()):
            arg = m.group(1).strip()
            lit = re.fullmatch(r'"([A-Za-z0-9_]+)"', arg)
            assert lit, f"{os.path.basename(path)}: getenv({arg}) is not a string literal"
            assert allowed.match(lit.group(1)), f"{os.path.basename(path)}: getenv(\"{lit.group(1)}\") is not on the allow-list"
            seen.add(lit.group(1))
    assert {"NERF_RCCL_LIB", "NERF_F16_TILES", "NERF_TRAIN_FORWARD"} <= seen      # the scan is not vacuous


def test_rccl_stand_in_builds_and_exports_the_bound_entry_points(tmp_path):
    """tests/stub_rccl.c (test-only; tests/test_gpu_multirank.py runs the library's multi-rank paths through it on a
    one-GPU box) exports exactly the six nccl* symbols csrc/comm_api.hip resolves with dlsym."""
    import subprocess
    out = tmp_path / "libstub_rccl.so"
    subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-I/opt/rocm/include", os.path.join(ROOT, "tests", "stub_rccl.c"),
                    "-o", str(out), "-L/opt/rocm/lib", "-lamdhip64", "-lrt"], check=True)
    syms = subprocess.run(["nm", "-D", "--defined-only", str(out)], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (nccl\w+)", syms))
    src = open(os.path.join(ROOT, "nerf_and_dietnerf_amd", "csrc", "comm_api.hip")).read()
    bound = set(re.findall(r'dlsym\(h, "(nccl\w+)"\)', src))
    assert len(bound) == 6 and exported == bound, (exported, bound)
