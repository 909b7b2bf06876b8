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
