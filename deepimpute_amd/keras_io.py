"""Keras-compatible persistence of a fitted MultiNet (reference deepimpute/multinet.py:105-124:
`model.to_json()` -> model.json, `model.save_weights()` -> model.h5, and `model_from_json` + `load_weights` on the way back).

* `model_json(...)`: the functional-model JSON Keras writes for the network build() defines (multinet.py:132-148): K
  InputLayers, per branch the Dense / Dropout chain, the softplus output Dense; layer names as Keras auto-generates them in
  build()'s creation order (input_1.., dense, dense_1, .., dropout, .., output layers last).  Our own metadata rides in an
  extra top-level key ("deepimpute_amd"), which model_from_json ignores.
* `write_weights_h5` / `read_weights_h5`: the HDF5 layout of Keras 2.x `save_weights` -- root attributes `layer_names`,
  `backend`, `keras_version`; one group per layer with attribute `weight_names`; datasets `<layer>/<layer>/kernel:0` and
  `bias:0` (float32, kernel [in][out]) -- through the HDF5 C library (libhdf5, ctypes) when the machine has one.
* `parse_model_json`: architecture list, input widths and the branch -> layer-name map back out of a Keras JSON (ours or
  one written by the reference), so that a model trained by the reference's Keras path can be loaded for predict().

STATUS: Keras / h5py are not installable here, so the files are validated structurally only (h5dump, round trip through
this module, tests/test_keras_io.py) -- never against Keras itself.  Without libhdf5 save() writes model.json + model.npz
only and says so; nothing else depends on this module.
"""
import ctypes as C
import ctypes.util
import json
import os

import numpy as np

KERAS_VERSION = "2.4.0"
_H5 = None


def _libhdf5():
    """The HDF5 C library, or None."""
    global _H5
    if _H5 is None:
        # Names the dynamic loader can try by itself first.  ctypes.util.find_library() comes LAST and only when none of them loads: it forks
        # ldconfig / gcc / ld, and a fork of a process that holds the 8 GB frame, the pinned bounce buffers and a HIP context took 0.8 s on a
        # quiet host and 15 s on a loaded one -- inside fit()'s save (round 4, profiles/r04_bench.json stages_s).
        import glob
        names = [os.environ.get("DIMN_LIBHDF5"), "libhdf5.so", "libhdf5_serial.so", "/opt/conda/lib/libhdf5.so",
                 "/usr/lib/x86_64-linux-gnu/libhdf5_serial.so"]
        names += sorted(glob.glob("/usr/lib/x86_64-linux-gnu/libhdf5*.so*") + glob.glob("/usr/lib64/libhdf5*.so*") + glob.glob("/opt/conda/lib/libhdf5.so*"))
        _H5 = False

        def attempt(name):
            global _H5
            try:
                lib = C.CDLL(name)
                lib.H5open()
                _H5 = _bind(lib)
                return True
            except (OSError, AttributeError):
                return False
        if not any(attempt(name) for name in names if name):
            found = ctypes.util.find_library("hdf5") or ctypes.util.find_library("hdf5_serial")
            if found:
                attempt(found)
    return _H5 or None


def available():
    return _libhdf5() is not None


def _bind(lib):
    hid, herr, sz = C.c_int64, C.c_int, C.c_size_t
    sig = {
        "H5Fcreate": (hid, [C.c_char_p, C.c_uint, hid, hid]), "H5Fopen": (hid, [C.c_char_p, C.c_uint, hid]), "H5Fclose": (herr, [hid]),
        "H5Gcreate2": (hid, [hid, C.c_char_p, hid, hid, hid]), "H5Gopen2": (hid, [hid, C.c_char_p, hid]), "H5Gclose": (herr, [hid]),
        "H5Screate_simple": (hid, [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]), "H5Screate": (hid, [C.c_int]), "H5Sclose": (herr, [hid]),
        "H5Sget_simple_extent_ndims": (C.c_int, [hid]), "H5Sget_simple_extent_dims": (C.c_int, [hid, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
        "H5Dcreate2": (hid, [hid, C.c_char_p, hid, hid, hid, hid, hid]), "H5Dopen2": (hid, [hid, C.c_char_p, hid]), "H5Dclose": (herr, [hid]),
        "H5Dwrite": (herr, [hid, hid, hid, hid, hid, C.c_void_p]), "H5Dread": (herr, [hid, hid, hid, hid, hid, C.c_void_p]),
        "H5Dget_space": (hid, [hid]),
        "H5Acreate2": (hid, [hid, C.c_char_p, hid, hid, hid, hid]), "H5Aopen": (hid, [hid, C.c_char_p, hid]), "H5Aclose": (herr, [hid]),
        "H5Awrite": (herr, [hid, hid, C.c_void_p]), "H5Aread": (herr, [hid, hid, C.c_void_p]), "H5Aget_type": (hid, [hid]), "H5Aget_space": (hid, [hid]),
        "H5Aexists": (C.c_int, [hid, C.c_char_p]),
        "H5Tcopy": (hid, [hid]), "H5Tset_size": (herr, [hid, sz]), "H5Tget_size": (sz, [hid]), "H5Tclose": (herr, [hid]), "H5Tis_variable_str": (C.c_int, [hid]),
        "H5Tset_strpad": (herr, [hid, C.c_int]),
        "H5Eset_auto2": (herr, [hid, C.c_void_p, C.c_void_p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    lib.H5Eset_auto2(0, None, None)                      # errors come back as negative ids; no stack traces on stderr
    lib.T_FLOAT = C.c_int64.in_dll(lib, "H5T_NATIVE_FLOAT_g").value
    lib.T_C_S1 = C.c_int64.in_dll(lib, "H5T_C_S1_g").value
    return lib


def _check(x, what):
    if x < 0:
        raise OSError("HDF5: %s failed" % what)
    return x


def _str_type(h5, size):
    t = _check(h5.H5Tcopy(h5.T_C_S1), "H5Tcopy")
    _check(h5.H5Tset_size(t, max(1, size)), "H5Tset_size")
    h5.H5Tset_strpad(t, 1)                               # H5T_STR_NULLPAD, what h5py writes for numpy 'S' arrays
    return t


def _write_str_attr(h5, obj, name, values, scalar=False):
    """A fixed-length string attribute: scalar (backend, keras_version) or a 1-D array (layer_names, weight_names)."""
    values = [v.encode("utf-8") if isinstance(v, str) else v for v in values]
    size = max([len(v) for v in values] + [1])
    t = _str_type(h5, size)
    if scalar:
        space = _check(h5.H5Screate(0), "H5Screate")     # H5S_SCALAR
    else:
        dims = (C.c_uint64 * 1)(len(values))
        space = _check(h5.H5Screate_simple(1, dims, None), "H5Screate_simple")
    a = _check(h5.H5Acreate2(obj, name.encode(), t, space, 0, 0), "H5Acreate2 " + name)
    buf = b"".join(v.ljust(size, b"\0") for v in values) or b"\0"
    if len(values):
        _check(h5.H5Awrite(a, t, buf), "H5Awrite " + name)
    h5.H5Aclose(a); h5.H5Sclose(space); h5.H5Tclose(t)


def _read_str_attr(h5, obj, name):
    if h5.H5Aexists(obj, name.encode()) <= 0:
        return []
    a = _check(h5.H5Aopen(obj, name.encode(), 0), "H5Aopen " + name)
    t, space = h5.H5Aget_type(a), h5.H5Aget_space(a)
    nd = h5.H5Sget_simple_extent_ndims(space)
    n = 1
    if nd == 1:
        dims = (C.c_uint64 * 1)()
        h5.H5Sget_simple_extent_dims(space, dims, None)
        n = int(dims[0])
    out = []
    if n:
        if h5.H5Tis_variable_str(t) > 0:                 # h5py >= 3 may write variable-length strings
            ptrs = (C.c_char_p * n)()
            _check(h5.H5Aread(a, t, ptrs), "H5Aread " + name)
            out = [(p or b"").decode("utf-8") for p in ptrs]
        else:
            size = int(h5.H5Tget_size(t))
            buf = C.create_string_buffer(size * n)
            _check(h5.H5Aread(a, t, buf), "H5Aread " + name)
            out = [buf.raw[i * size:(i + 1) * size].split(b"\0")[0].decode("utf-8") for i in range(n)]
    h5.H5Tclose(t); h5.H5Sclose(space); h5.H5Aclose(a)
    return out


# ----------------------------------------------------------------------------------------------- layer naming
def _suffix(base, i):
    return base if i == 0 else "%s_%d" % (base, i)


def layer_names(K, layers):
    """Keras' automatic names in build()'s creation order (multinet.py:132-146): inputs input_1..input_K; then for every
    architecture entry, one layer per branch; the K output Dense layers last.  -> (inputs, hidden[l][k], dropout[l][k] or None, outputs[k]);
    hidden[0] is None for a leading (0, _, rate) entry = a Dropout layer before the first Dense layer."""
    inputs = ["input_%d" % (k + 1) for k in range(K)]
    dense_i = drop_i = 0
    hidden, drops = [], []
    for units, _, rate in layers:
        if units == 0:                                   # a Dropout layer before the first Dense layer: no Dense of its own
            hidden.append(None)
        else:
            hidden.append([_suffix("dense", dense_i + k) for k in range(K)])
            dense_i += K
        if rate > 0:
            drops.append([_suffix("dropout", drop_i + k) for k in range(K)])
            drop_i += K
        else:
            drops.append(None)
    outputs = [_suffix("dense", dense_i + k) for k in range(K)]
    return inputs, hidden, drops, outputs


def model_json(inputdims, layers, out_dim, seed, extra=None):
    """The Keras functional-model config (model.to_json()) of the network build() defines."""
    K = len(inputdims)
    inputs, hidden, drops, outputs = layer_names(K, layers)

    def dense(name, units, act, src):
        return {"class_name": "Dense", "name": name, "inbound_nodes": [[[src, 0, 0, {}]]],
                "config": {"name": name, "trainable": True, "dtype": "float32", "units": int(units), "activation": act, "use_bias": True,
                           "kernel_initializer": {"class_name": "GlorotUniform", "config": {"seed": None}},
                           "bias_initializer": {"class_name": "Zeros", "config": {}}, "kernel_regularizer": None, "bias_regularizer": None,
                           "activity_regularizer": None, "kernel_constraint": None, "bias_constraint": None}}
    cfg_layers = [{"class_name": "InputLayer", "name": nm, "inbound_nodes": [],
                   "config": {"batch_input_shape": [None, int(d)], "dtype": "float32", "sparse": False, "ragged": False, "name": nm}}
                  for nm, d in zip(inputs, inputdims)]
    prev = list(inputs)
    for l, (units, act, rate) in enumerate(layers):
        if hidden[l] is not None:
            for k in range(K):
                cfg_layers.append(dense(hidden[l][k], units, act, prev[k]))
            prev = list(hidden[l])
        if drops[l] is not None:
            for k in range(K):
                cfg_layers.append({"class_name": "Dropout", "name": drops[l][k], "inbound_nodes": [[[prev[k], 0, 0, {}]]],
                                   "config": {"name": drops[l][k], "trainable": True, "dtype": "float32", "rate": float(rate), "noise_shape": None,
                                              "seed": None if seed is None else int(seed)}})
            prev = list(drops[l])
    for k in range(K):
        cfg_layers.append(dense(outputs[k], out_dim, "softplus", prev[k]))
    doc = {"class_name": "Functional", "config": {"name": "model", "layers": cfg_layers, "input_layers": [[nm, 0, 0] for nm in inputs],
                                                  "output_layers": [[nm, 0, 0] for nm in outputs]},
           "keras_version": KERAS_VERSION, "backend": "tensorflow"}
    if extra:
        doc["deepimpute_amd"] = extra
    return doc


def parse_model_json(doc):
    """(inputdims, architecture list, out_dim, dense layer names per branch [k][l]) of a Keras functional JSON of the family
    build() produces (K disjoint Input -> (Dense [-> Dropout])* -> Dense(softplus) chains)."""
    cfg = doc["config"]
    by_name = {l["name"]: l for l in cfg["layers"]}
    inputs = [x[0] for x in cfg["input_layers"]]
    outputs = [x[0] for x in cfg["output_layers"]]
    chains = []
    for out in outputs:
        chain, name = [], out
        while True:
            layer = by_name[name]
            chain.append(layer)
            if layer["class_name"] == "InputLayer":
                break
            name = layer["inbound_nodes"][0][0][0]
        chains.append(chain[::-1])
    chains.sort(key=lambda ch: inputs.index(ch[0]["name"]))          # branch k = the k-th model input (multinet.py:231-235 feeds them in order)
    inputdims = [int(ch[0]["config"]["batch_input_shape"][1]) for ch in chains]
    arch, names = None, []
    for ch in chains:
        a, dn = [], []
        for layer in ch[1:]:
            if layer["class_name"] == "Dense":
                a.append({"type": "dense", "neurons": int(layer["config"]["units"]), "activation": layer["config"]["activation"]})
                dn.append(layer["name"])
            elif layer["class_name"] == "Dropout":
                a.append({"type": "dropout", "rate": float(layer["config"]["rate"])})
            else:
                raise NotImplementedError("layer class %r in model.json" % layer["class_name"])
        out_dim = a.pop()["neurons"]                                 # the output Dense is not part of `architecture`
        if arch is not None and a != arch:
            raise NotImplementedError("the branches of model.json differ")
        arch = a
        names.append(dn)
    return inputdims, arch, out_dim, names


# ----------------------------------------------------------------------------------------------- weights
def write_weights_h5(path, layer_order, weights):
    """Keras 2.x save_weights layout.  layer_order: every layer name of the model in order; weights: {dense name: (kernel [in][out], bias)}."""
    h5 = _libhdf5()
    if h5 is None:
        raise OSError("no HDF5 library on this machine")
    f = _check(h5.H5Fcreate(path.encode(), 2, 0, 0), "H5Fcreate " + path)        # H5F_ACC_TRUNC
    try:
        _write_str_attr(h5, f, "layer_names", layer_order)
        _write_str_attr(h5, f, "backend", ["tensorflow"], scalar=True)
        _write_str_attr(h5, f, "keras_version", [KERAS_VERSION], scalar=True)
        for name in layer_order:
            g = _check(h5.H5Gcreate2(f, name.encode(), 0, 0, 0), "H5Gcreate2 " + name)
            arrays = weights.get(name)
            wn = ["%s/kernel:0" % name, "%s/bias:0" % name] if arrays is not None else []
            _write_str_attr(h5, g, "weight_names", wn)
            if arrays is not None:
                inner = _check(h5.H5Gcreate2(g, name.encode(), 0, 0, 0), "H5Gcreate2 inner " + name)
                for ds_name, arr in zip(("kernel:0", "bias:0"), arrays):
                    arr = np.ascontiguousarray(arr, np.float32)
                    dims = (C.c_uint64 * arr.ndim)(*arr.shape)
                    space = _check(h5.H5Screate_simple(arr.ndim, dims, None), "H5Screate_simple")
                    d = _check(h5.H5Dcreate2(inner, ds_name.encode(), h5.T_FLOAT, space, 0, 0, 0), "H5Dcreate2 " + ds_name)
                    _check(h5.H5Dwrite(d, h5.T_FLOAT, 0, 0, 0, arr.ctypes.data_as(C.c_void_p)), "H5Dwrite " + ds_name)
                    h5.H5Dclose(d); h5.H5Sclose(space)
                h5.H5Gclose(inner)
            h5.H5Gclose(g)
    finally:
        h5.H5Fclose(f)


def read_weights_h5(path, only=None):
    """{layer name: [arrays in weight_names order]} of a Keras save_weights file (top-level layout, or under /model_weights
    as model.save() nests it); `only`: the layer names wanted (a rank of a sharded job reads its own sub-nets)."""
    h5 = _libhdf5()
    if h5 is None:
        raise OSError("no HDF5 library on this machine")
    f = _check(h5.H5Fopen(path.encode(), 0, 0), "H5Fopen " + path)               # H5F_ACC_RDONLY
    try:
        root = f
        if not _read_str_attr(h5, f, "layer_names"):
            root = h5.H5Gopen2(f, b"model_weights", 0)
            if root < 0:
                raise OSError("%s: no layer_names attribute (not a Keras weights file)" % path)
        out = {}
        for name in _read_str_attr(h5, root, "layer_names"):
            if only is not None and name not in only:
                continue
            g = _check(h5.H5Gopen2(root, name.encode(), 0), "H5Gopen2 " + name)
            arrays = []
            for wn in _read_str_attr(h5, g, "weight_names"):
                d = _check(h5.H5Dopen2(g, wn.encode(), 0), "H5Dopen2 " + wn)
                space = h5.H5Dget_space(d)
                nd = h5.H5Sget_simple_extent_ndims(space)
                dims = (C.c_uint64 * max(nd, 1))()
                h5.H5Sget_simple_extent_dims(space, dims, None)
                arr = np.empty([int(x) for x in dims[:nd]], np.float32)
                _check(h5.H5Dread(d, h5.T_FLOAT, 0, 0, 0, arr.ctypes.data_as(C.c_void_p)), "H5Dread " + wn)
                arrays.append(arr)
                h5.H5Sclose(space); h5.H5Dclose(d)
            h5.H5Gclose(g)
            out[name] = arrays
        if root != f:
            h5.H5Gclose(root)
        return out
    finally:
        h5.H5Fclose(f)
