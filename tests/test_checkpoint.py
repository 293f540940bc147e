"""TensorFlow tensor-bundle checkpoints: the reader is pinned by the reference's own checkpoint fixtures
(tests/data/model-checkpoints/{las,ds}.ckpt, written by TensorFlow - every table block, tensor and
string checksum in them verifies), the writer by reading its output back and by comparing the object
graph it emits with the one TensorFlow wrote for the same model."""
import os

import numpy as np
import pytest
import torch

from speech_recognition_amd import checkpoint as ck

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_fixtures")
LAS_CKPT, DS_CKPT = os.path.join(FIX, "las.ckpt"), os.path.join(FIX, "ds.ckpt")


def _las_mini(device="cuda"):
    from speech_recognition_amd.configs import get_model_config
    return get_model_config(os.path.join(FIX, "las_mini_for_test.yml")).create_model(device=device)


def _ds_mini(device="cuda"):
    from speech_recognition_amd.configs import get_model_config
    return get_model_config(os.path.join(FIX, "deepspeech_mini_for_test.yml")).create_model(device=device)


@pytest.mark.parametrize("prefix,n_vars", [(LAS_CKPT, 30), (DS_CKPT, 74)])
def test_reference_checkpoints_read_with_all_checksums(prefix, n_vars):
    index = ck.read_index(prefix + ".index", verify=True)
    assert ck.OBJECT_GRAPH_KEY in index and "" in index
    variables = ck.load_variables(prefix, verify=True)
    assert len(variables) == n_vars
    assert all(v.dtype == np.float32 and np.isfinite(v).all() for v in variables.values())


@pytest.mark.parametrize("make,prefix", [(_las_mini, LAS_CKPT), (_ds_mini, DS_CKPT)])
def test_variable_names_and_shapes_equal_the_reference_checkpoint(make, prefix):
    """The layout contract of SURVEY.md 8b: Keras variable paths and shapes (kernel [Din, G*H], HWIO convs,
    ...) - checked against what TensorFlow actually wrote for the reference's mini models."""
    model = make("cpu")                                      # shapes only: nothing is allocated or computed
    *trainable, non_trainable = model.param_shapes(80, 3)   # dicts name -> shape; the last holds the BN moving statistics
    shapes = {k: tuple(v) for d in trainable + [non_trainable] for k, v in d.items()}
    variables = ck.load_variables(prefix)
    assert set(shapes) == set(variables)
    for k, shp in shapes.items():
        assert tuple(variables[k].shape) == shp, k
    assert set(non_trainable) == {k for k in variables if k.endswith("moving_mean") or k.endswith("moving_variance")}


def test_corruption_is_detected(tmp_path):
    import shutil
    for ext in (".index", ".data-00000-of-00001"):
        shutil.copy(LAS_CKPT + ext, tmp_path / ("c.ckpt" + ext))
    p = str(tmp_path / "c.ckpt")
    with open(p + ".data-00000-of-00001", "r+b") as f:
        f.seek(1000)
        b = f.read(1)
        f.seek(1000)
        f.write(bytes([b[0] ^ 1]))
    with pytest.raises(ValueError, match="checksum"):
        ck.read_bundle(p)
    assert len(ck.read_bundle(p, verify=False)) == 30
    with open(p + ".index", "r+b") as f:
        f.seek(40)
        f.write(b"\xff")
    with pytest.raises(ValueError):
        ck.read_index(p + ".index")
    with pytest.raises(ValueError, match="magic"):
        (tmp_path / "x.index").write_bytes(b"\0" * 64)
        ck.read_index(str(tmp_path / "x.index"))


def test_writer_round_trip_and_object_graph_matches_tensorflows(tmp_path):
    variables = ck.load_variables(LAS_CKPT)
    out = str(tmp_path / "model.ckpt")
    ck.save_variables(out, variables)
    back = ck.load_variables(out, verify=True)
    assert set(back) == set(variables)
    for k in variables:
        np.testing.assert_array_equal(back[k], variables[k])
    mine, theirs = ck.read_object_graph(out), ck.read_object_graph(LAS_CKPT)

    def resolve(graph, path):
        node = 0
        for part in path.split("/"):
            node = graph[node]["children"][part]
        return graph[node]["attributes"]

    for name in variables:                                   # same walk from the root reaches the same checkpoint key
        a, b = resolve(mine, name), resolve(theirs, name)
        assert [x["checkpoint_key"] for x in a] == [x["checkpoint_key"] for x in b] == [name + "/.ATTRIBUTES/VARIABLE_VALUE"]
        assert a[0]["name"] == b[0]["name"] == "VARIABLE_VALUE"
    # odd dtypes / shapes and key ordering survive too
    odd = {"Zeta/x": np.arange(6, dtype=np.int32).reshape(2, 3), "alpha": np.float32(3.5), "m/empty": np.zeros((0, 4), np.float32),
           "m/i64": np.array([2 ** 40, -1], np.int64)}
    ck.write_bundle(str(tmp_path / "odd"), odd)
    got = ck.read_bundle(str(tmp_path / "odd"))
    assert set(got) == set(odd)
    for k in odd:
        np.testing.assert_array_equal(got[k], odd[k])
        assert got[k].dtype == np.asarray(odd[k]).dtype


@pytest.mark.gpu
@pytest.mark.parametrize("make,prefix,shape", [(_las_mini, LAS_CKPT, (2, 120, 80, 3)), (_ds_mini, DS_CKPT, (2, 200, 80, 3))])
def test_reference_checkpoint_loads_runs_and_saves_back(make, prefix, shape, tmp_path):
    """run/train.py:152-154 --pretrained-model-path with the reference's files; then save_weights and compare."""
    model = make()
    model.build(80, 3)
    model.load_weights(prefix)
    ref = ck.load_variables(prefix)
    now = model.state_dict()
    assert set(now) == set(ref)
    for k in ref:
        assert torch.equal(now[k], torch.from_numpy(ref[k])), k
    g = torch.Generator().manual_seed(0)
    audio = torch.randn(*shape, generator=g).cuda()
    if "las" in prefix:
        tokens = torch.randint(1, 3000, (2, 5), generator=g, dtype=torch.int32).cuda()
        out = model((audio, tokens), training=False)
        assert tuple(out.shape) == (2, 5, 3000)
    else:
        out = model(audio, training=False)
        assert out.shape[0] == 2 and out.shape[2] == 120
    assert torch.isfinite(out).all()
    path = str(tmp_path / "again.ckpt")
    model.save_weights(path)
    again = ck.load_variables(path)
    for k in ref:
        np.testing.assert_array_equal(again[k], ref[k])
    other = _ds_mini() if "las" in prefix else _las_mini()          # the wrong architecture must be refused
    other.build(80, 3)
    with pytest.raises(ValueError, match="lacks|shape"):
        other.load_weights(prefix)
