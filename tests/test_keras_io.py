"""Keras-compatible persistence (reference deepimpute/multinet.py:105-124: model.json + model.h5).  Structural checks only:
Keras / h5py are not installable here, so the HDF5 layout is checked through this module's own reader and h5dump."""
import json
import os
import shutil
import subprocess

import numpy as np
import pandas as pd
import pytest

from helpers import multinet_with

from deepimpute_amd import keras_io
from deepimpute_amd.multinet import MultiNet

needs_hdf5 = pytest.mark.skipif(not keras_io.available(), reason="no HDF5 library on this machine")


def test_layer_names_follow_keras_creation_order():
    # build() (multinet.py:132-146): all inputs, then per architecture entry one layer per branch, outputs last;
    # Keras numbers layers of a class in creation order: dense, dense_1, ...
    ins, hid, drops, outs = keras_io.layer_names(3, [(32, "relu", 0.2), (16, "tanh", 0.0)])
    assert ins == ["input_1", "input_2", "input_3"]
    assert hid == [["dense", "dense_1", "dense_2"], ["dense_3", "dense_4", "dense_5"]]
    assert drops == [["dropout", "dropout_1", "dropout_2"], None]
    assert outs == ["dense_6", "dense_7", "dense_8"]


def test_input_dropout_is_a_dropout_layer_between_the_input_and_the_first_dense():
    """A Dropout layer BEFORE the first Dense layer (multinet.py:139-141 builds it like any other): the leading (0, _, rate) entry of the
    engines' layer list; in model.json a Dropout node fed by the InputLayer, numbered first among the Dropout layers."""
    layers = [(0, "linear", 0.15), (24, "tanh", 0.1), (16, "relu", 0.0)]
    ins, hid, drops, outs = keras_io.layer_names(2, layers)
    assert hid == [None, ["dense", "dense_1"], ["dense_2", "dense_3"]] and outs == ["dense_4", "dense_5"]
    assert drops == [["dropout", "dropout_1"], ["dropout_2", "dropout_3"], None]
    doc = json.loads(json.dumps(keras_io.model_json([7, 9], layers, 12, 3)))
    by_name = {l["name"]: l for l in doc["config"]["layers"]}
    assert by_name["dropout"]["inbound_nodes"][0][0][0] == "input_1" and by_name["dense"]["inbound_nodes"][0][0][0] == "dropout"
    inputdims, arch, out_dim, names = keras_io.parse_model_json(doc)
    assert inputdims == [7, 9] and out_dim == 12 and names == [["dense", "dense_2", "dense_4"], ["dense_1", "dense_3", "dense_5"]]
    assert arch == [{"type": "dropout", "rate": 0.15}, {"type": "dense", "neurons": 24, "activation": "tanh"}, {"type": "dropout", "rate": 0.1},
                    {"type": "dense", "neurons": 16, "activation": "relu"}]
    from deepimpute_amd.multinet import _parse_architecture
    assert _parse_architecture(arch) == [tuple(l) for l in layers]


@pytest.mark.parametrize("layers", [[(256, "relu", 0.2)], [(64, "relu", 0.3), (32, "sigmoid", 0.0), (16, "linear", 0.1)]])
def test_model_json_round_trip(layers):
    dims = [7, 11, 5, 9]
    doc = json.loads(json.dumps(keras_io.model_json(dims, layers, 48, 1234, {"format": "x"})))
    assert doc["class_name"] == "Functional" and doc["deepimpute_amd"] == {"format": "x"}
    cfg = doc["config"]
    assert [x[0] for x in cfg["input_layers"]] == ["input_%d" % (k + 1) for k in range(4)]
    inputdims, arch, out_dim, names = keras_io.parse_model_json(doc)
    assert inputdims == dims and out_dim == 48
    want = []
    for units, act, rate in layers:
        want.append({"type": "dense", "neurons": units, "activation": act})
        if rate > 0:
            want.append({"type": "dropout", "rate": rate})
    assert arch == want
    _, hid, _, outs = keras_io.layer_names(4, layers)
    assert names == [[h[k] for h in hid] + [outs[k]] for k in range(4)]
    # the output layers are softplus Dense layers fed by the last hidden (or dropout) layer of the same branch
    by_name = {l["name"]: l for l in cfg["layers"]}
    for k in range(4):
        assert by_name[outs[k]]["config"]["activation"] == "softplus" and by_name[outs[k]]["config"]["units"] == 48
    # the inputs may come in any order in `layers`; branches follow `input_layers`
    cfg["layers"] = cfg["layers"][::-1]
    assert keras_io.parse_model_json(doc)[3] == names


def _weights(dims, layers, out_dim, seed=0):
    rng = np.random.default_rng(seed)
    K = len(dims)
    ins, hid, drops, outs = keras_io.layer_names(K, layers)
    order = list(ins)
    for l in range(len(layers)):
        order += (hid[l] or []) + (drops[l] or [])
    order += outs
    weights = {}
    for k in range(K):
        fan_in = dims[k]
        for l, (units, _, _) in enumerate(layers):
            if hid[l] is None:                                   # a leading input-dropout entry: no Dense, no weights
                continue
            weights[hid[l][k]] = (rng.standard_normal((fan_in, units)).astype(np.float32), rng.standard_normal(units).astype(np.float32))
            fan_in = units
        weights[outs[k]] = (rng.standard_normal((fan_in, out_dim)).astype(np.float32), rng.standard_normal(out_dim).astype(np.float32))
    return order, weights


@needs_hdf5
def test_weights_h5_round_trip_and_layout(tmp_path):
    layers = [(32, "relu", 0.2), (16, "tanh", 0.0)]
    order, weights = _weights([7, 300, 5], layers, 24)
    path = str(tmp_path / "model.h5")
    keras_io.write_weights_h5(path, order, weights)
    back = keras_io.read_weights_h5(path)
    assert list(back) == order                                       # layer_names keeps the model's layer order
    for name in order:
        if name in weights:
            assert len(back[name]) == 2
            for a, b in zip(weights[name], back[name]):
                assert a.dtype == b.dtype == np.float32 and a.shape == b.shape and np.array_equal(a, b)
        else:
            assert back[name] == []                                  # inputs, dropout: a group with an empty weight_names
    some = keras_io.read_weights_h5(path, only={"dense_4"})
    assert list(some) == ["dense_4"] and np.array_equal(some["dense_4"][0], weights["dense_4"][0])

    h5dump = shutil.which("h5dump") or ("/opt/conda/bin/h5dump" if os.path.exists("/opt/conda/bin/h5dump") else None)
    if h5dump is None:
        return
    # the object tree Keras 2.x save_weights produces: /<layer>/<layer>/{kernel:0,bias:0}, attributes on / and on each layer group
    text = subprocess.run([h5dump, "-n", "1", path], capture_output=True, text=True, check=True).stdout
    lines = [l.split() for l in text.splitlines()]
    objs = {tuple(l[:2]) for l in lines if len(l) >= 2}
    for name in weights:
        assert ("group", "/%s/%s" % (name, name)) in objs
        assert ("dataset", "/%s/%s/kernel:0" % (name, name)) in objs and ("dataset", "/%s/%s/bias:0" % (name, name)) in objs
        assert ("attribute", "/%s/weight_names" % name) in objs
    for attr in ("layer_names", "backend", "keras_version"):
        assert ("attribute", "/" + attr) in objs
    text = subprocess.run([h5dump, "-a", "/dense_3/weight_names", "-a", "/backend", path], capture_output=True, text=True, check=True).stdout
    assert '"dense_3/kernel:0"' in text and '"dense_3/bias:0' in text and '"tensorflow"' in text      # NULLPAD strings, as numpy 'S' arrays are stored


@needs_hdf5
def test_reads_the_nested_layout_of_model_save(tmp_path):
    """`read_weights_h5` also takes files where the weights sit under /model_weights (Keras model.save())."""
    h5 = keras_io._libhdf5()
    order, weights = _weights([4], [(8, "relu", 0.0)], 3)
    flat = str(tmp_path / "flat.h5")
    keras_io.write_weights_h5(flat, order, weights)
    # build the nested file with the same primitives: / -> model_weights -> (the layout of `flat`)
    nested = str(tmp_path / "nested.h5")
    f = h5.H5Fcreate(nested.encode(), 2, 0, 0)
    g = h5.H5Gcreate2(f, b"model_weights", 0, 0, 0)
    keras_io._write_str_attr(h5, g, "layer_names", order)
    for name in order:
        lg = h5.H5Gcreate2(g, name.encode(), 0, 0, 0)
        keras_io._write_str_attr(h5, lg, "weight_names", ["%s/kernel:0" % name] if name in weights else [])
        if name in weights:
            inner = h5.H5Gcreate2(lg, name.encode(), 0, 0, 0)
            arr = weights[name][0]
            import ctypes as C
            dims = (C.c_uint64 * 2)(*arr.shape)
            sp = h5.H5Screate_simple(2, dims, None)
            d = h5.H5Dcreate2(inner, b"kernel:0", h5.T_FLOAT, sp, 0, 0, 0)
            h5.H5Dwrite(d, h5.T_FLOAT, 0, 0, 0, arr.ctypes.data_as(C.c_void_p))
            h5.H5Dclose(d); h5.H5Sclose(sp); h5.H5Gclose(inner)
        h5.H5Gclose(lg)
    h5.H5Gclose(g); h5.H5Fclose(f)
    back = keras_io.read_weights_h5(nested)
    assert np.array_equal(back["dense"][0], weights["dense"][0]) and len(back["dense"]) == 1


def test_not_a_weights_file_is_loud(tmp_path):
    if not keras_io.available():
        pytest.skip("no HDF5 library")
    bad = tmp_path / "x.h5"
    bad.write_bytes(b"not hdf5")
    with pytest.raises(OSError):
        keras_io.read_weights_h5(str(bad))


class _HoldEngine:
    """An engine that only stores weights (the persistence path needs nothing else)."""
    def __init__(self, D, hidden, out_dim, **kw):
        self.D, self.K, self.kw = list(D), len(D), kw
        self.layers = hidden if isinstance(hidden, list) else [(hidden, "relu", kw.get("dropout_rate", 0.0))]
        self.O, self.w = out_dim, {}

    @classmethod
    def general(cls, D, layers, out_dim, **kw):
        return cls(D, list(layers), out_dim, **kw)

    def set_weights(self, k, *arrays):
        self.w[k] = [np.array(a) for a in arrays]

    def get_weights(self, k):
        return self.w[k]

    def close(self):
        pass


@pytest.mark.parametrize("fmt", ["h5", "npz", "both"])
@pytest.mark.parametrize("arch", [None, [{"type": "dense", "neurons": 24, "activation": "tanh"}, {"type": "dropout", "rate": 0.1},
                                         {"type": "dense", "neurons": 12, "activation": "relu"}],
                                  [{"type": "dropout", "rate": 0.15}, {"type": "dense", "neurons": 10, "activation": "gelu"}]])
def test_multinet_save_load(tmp_path, monkeypatch, fmt, arch):
    if fmt != "npz" and not keras_io.available():
        pytest.skip("no HDF5 library")
    monkeypatch.setenv("DIMN_MODEL_FORMAT", fmt)
    dims = [6, 9, 4]
    net = multinet_with(_HoldEngine, sub_outputdim=20, output_prefix=str(tmp_path), architecture=arch, verbose=0,
                   loss="mean_squared_error" if arch else "wMSE", batch_size=32)
    net.predictors = [["g%d" % j for j in range(d)] for d in dims]
    eng = net.build(dims)
    layers = eng.layers
    _, weights = _weights(dims, layers, 20, seed=5)
    _, hid, _, outs = keras_io.layer_names(3, layers)
    hid = [h for h in hid if h is not None]                      # (a leading input-dropout entry has no Dense of its own)
    for k in range(3):
        eng.set_weights(k, *[a for name in [h[k] for h in hid] + [outs[k]] for a in weights[name]])
    net.save(eng)
    files = set(os.listdir(str(tmp_path)))
    assert files == {"model.json"} | {"h5": {"model.h5"}, "npz": {"model.npz"}, "both": {"model.h5", "model.npz"}}[fmt]
    doc = json.load(open(str(tmp_path / "model.json")))
    assert doc["class_name"] == "Functional" and doc["deepimpute_amd"]["weights"] == fmt       # model_from_json ignores the extra key

    fresh = multinet_with(_HoldEngine, output_prefix=str(tmp_path), verbose=0)
    got = fresh.load()
    assert fresh.sub_outputdim == 20 and fresh.NN_parameters["batch_size"] == 32
    assert got.D == dims and [tuple(l) for l in got.layers] == [tuple(l) for l in layers]
    for k in range(3):
        assert len(got.w[k]) == len(eng.w[k])
        for a, b in zip(got.w[k], eng.w[k]):
            assert np.array_equal(a, b)

    if fmt == "h5":
        # the reference's own pair: a Keras model.json WITHOUT our metadata + model.h5
        del doc["deepimpute_amd"]
        json.dump(doc, open(str(tmp_path / "model.json"), "w"))
        plain = multinet_with(_HoldEngine, output_prefix=str(tmp_path), verbose=0, batch_size=32,
                         loss="mean_squared_error" if arch else "wMSE")
        got = plain.load()
        assert plain.sub_outputdim == 20 and got.D == dims
        for k in range(3):
            for a, b in zip(got.w[k], eng.w[k]):
                assert np.array_equal(a, b)


def test_stale_weights_of_an_older_fit_are_removed(tmp_path, monkeypatch):
    if not keras_io.available():
        pytest.skip("no HDF5 library")
    dims = [5, 5]
    for fmt in ("npz", "h5"):
        monkeypatch.setenv("DIMN_MODEL_FORMAT", fmt)
        net = multinet_with(_HoldEngine, sub_outputdim=8, output_prefix=str(tmp_path), verbose=0)
        net.predictors = [list(range(d)) for d in dims]
        eng = net.build(dims)
        _, weights = _weights(dims, eng.layers, 8, seed=1)
        _, hid, _, outs = keras_io.layer_names(2, eng.layers)
        for k in range(2):
            eng.set_weights(k, *[a for name in (hid[0][k], outs[k]) for a in weights[name]])
        net.save(eng)
    assert sorted(os.listdir(str(tmp_path))) == ["model.h5", "model.json"]
