"""nerf_and_dietnerf_amd.keras_h5: the pure-Python reader of Keras .h5 weight files."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_CKPT = ("/root/reference/Results/50px_alexander_71pics_sphere_nerf_save_dir_4/saved_weights/"
            "NeRF_model_epoch_095.h5")


def test_tiny_keras_layout():
    from nerf_and_dietnerf_amd import keras_h5
    exp = np.load(os.path.join(ROOT, "tests", "golden", "tiny_keras_layout.npz"))
    got = keras_h5.read_keras_weights(os.path.join(ROOT, "tests", "golden", "tiny_keras_layout.h5"))
    assert list(got) == ["model", "model_1"]
    for g in ("model", "model_1"):
        assert len(got[g]) == 24                       # 12 layers x (kernel, bias), natural layer order
        for li in range(12):
            np.testing.assert_array_equal(got[g][2 * li], exp[f"{g}/{li}/kernel"])
            np.testing.assert_array_equal(got[g][2 * li + 1], exp[f"{g}/{li}/bias"])
    c, f = keras_h5.load_nerf_checkpoint(os.path.join(ROOT, "tests", "golden", "tiny_keras_layout.h5"))
    assert c.size == f.size == sum(exp[f"model/{li}/kernel"].size + 5 for li in range(12))


def test_rejects_non_hdf5(tmp_path):
    from nerf_and_dietnerf_amd import keras_h5
    p = tmp_path / "x.h5"
    p.write_bytes(b"not hdf5 at all" * 10)
    with pytest.raises(ValueError, match="not an HDF5"):
        keras_h5.read_keras_weights(str(p))


@pytest.mark.skipif(not os.path.exists(REF_CKPT), reason="reference tree not present (GPU box)")
def test_reference_checkpoint_matches_fixture(golden_ckpt):
    """The shipped epoch-95 checkpoint read without h5py == the h5py-extracted fixture, bit for bit."""
    from nerf_and_dietnerf_amd import keras_h5
    c, f = keras_h5.load_nerf_checkpoint(REF_CKPT)
    np.testing.assert_array_equal(c, golden_ckpt["blob_coarse"])
    np.testing.assert_array_equal(f, golden_ckpt["blob_fine"])


def test_writer_roundtrip_and_h5py_crosscheck(golden_ckpt, tmp_path):
    """save_nerf_checkpoint -> load_nerf_checkpoint is the identity, the layout is the reference's (groups model /
    model_1 / top_level_model_weights, dense .. dense_21), and -- when the build container's second interpreter with
    h5py is present -- libhdf5 itself reads the file back bit-identically, attributes included."""
    import os
    import subprocess
    from nerf_and_dietnerf_amd import keras_h5 as K
    path = str(tmp_path / "NeRF_model_epoch_001.h5")
    K.save_nerf_checkpoint(path, golden_ckpt["blob_coarse"], golden_ckpt["blob_fine"])
    c, f = K.load_nerf_checkpoint(path)
    np.testing.assert_array_equal(c, golden_ckpt["blob_coarse"])
    np.testing.assert_array_equal(f, golden_ckpt["blob_fine"])
    models = K.read_keras_weights(path)
    assert list(models) == ["model", "model_1"] and len(models["model"]) == 22
    # coarse-only checkpoint and the xyz-only network's 12 layers
    K.save_nerf_checkpoint(path, np.arange(577028, dtype=np.float32), None, n_angles=0)
    c0, f0 = K.load_nerf_checkpoint(path)
    assert f0 is None and c0.size == 577028 and c0[-1] == 577027
    conda = "/opt/conda/bin/python3.9"
    if not os.path.exists(conda):
        pytest.skip("no h5py interpreter here (GPU box): libhdf5 cross-check runs in the build container")
    K.save_nerf_checkpoint(path, golden_ckpt["blob_coarse"], golden_ckpt["blob_fine"])
    code = (
        "import h5py, numpy as np, sys\n"
        "f = h5py.File(sys.argv[1], 'r')\n"
        "assert sorted(f.keys()) == ['model', 'model_1', 'top_level_model_weights']\n"
        "assert [n.decode() for n in f.attrs['layer_names']] == ['model', 'model_1']\n"
        "assert f.attrs['backend'] == b'tensorflow' and f.attrs['keras_version'] == b'2.7.0'\n"
        "assert f['top_level_model_weights'].attrs['weight_names'].shape == (0,)\n"
        "ref = np.load(sys.argv[2])\n"
        "for g, k in (('model', 'blob_coarse'), ('model_1', 'blob_fine')):\n"
        "    wn = [n.decode() for n in f[g].attrs['weight_names']]\n"
        "    blob = np.concatenate([np.asarray(f[g][n], np.float32).ravel() for n in wn])\n"
        "    assert np.array_equal(blob, ref[k]), g\n"
        "assert f['model_1/dense_21/bias:0'].shape == (1,)\n"
        "print('ok')\n")
    fixture = os.path.join(os.path.dirname(__file__), "golden", "alexander50_epoch095.npz")
    r = subprocess.run([conda, "-W", "ignore", "-c", code, path, fixture], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]
