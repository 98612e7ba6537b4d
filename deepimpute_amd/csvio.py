"""The CSV edges of the CLI (reference deepimpute/deepImpute.py:13 `pd.read_csv(inputFile, index_col=0)` and :35
`imputed.to_csv(output)`) on libdimn's multi-threaded host reader / writer (include/dimn.h, csrc/dimn_csv.h).

At 50k cells x 20k genes the two pandas calls take minutes around seconds of GPU work.  The native reader takes exactly
the input the tool is specified for -- a rectangular matrix of raw integer counts with unquoted labels -- and returns the
frame pandas would (int64 values; an all-integer label column becomes an integer index, as pandas infers it); for any
other content (decimals, empty or quoted fields, ragged rows, labels that would need quoting) these functions call
pandas themselves: a change of speed, never of result.
"""
import ctypes as C
import re

import numpy as np
import pandas as pd

_INT = re.compile(r"^[+-]?[0-9]+$")


def _fns():
    from . import _lib
    return _lib.load()


# pandas' default na_values (pandas.io.parsers: STR_NA_VALUES): an index label among them becomes NaN in pd.read_csv
_PANDAS_NA = frozenset(["", "#N/A", "#N/A N/A", "#NA", "-1.#IND", "-1.#QNAN", "-NaN", "-nan", "1.#IND", "1.#QNAN", "<NA>", "N/A", "NA",
                        "NULL", "NaN", "None", "n/a", "nan", "null"])


def read_csv(path):
    """pd.read_csv(path, index_col=0)."""
    try:
        fns = _fns()
    except (ImportError, OSError):
        return pd.read_csv(path, index_col=0)
    n, g, lb = C.c_int64(), C.c_int64(), C.c_int64()
    if fns["csv_scan"](str(path).encode(), C.byref(n), C.byref(g), C.byref(lb)) != 0:
        return pd.read_csv(path, index_col=0)
    values = np.empty((n.value, g.value), np.int64)
    labels = C.create_string_buffer(lb.value)
    if fns["csv_read"](str(path).encode(), n.value, g.value, values.ctypes.data_as(C.POINTER(C.c_int64)), labels, lb.value) != 0:
        return pd.read_csv(path, index_col=0)
    try:
        parts = labels.raw.split(b"\0")[:1 + g.value + n.value]
        text = [p.decode("utf-8") for p in parts]
    except UnicodeDecodeError:
        return pd.read_csv(path, index_col=0)
    name, cols, rows = text[0], text[1:1 + g.value], text[1 + g.value:]
    if len(set(cols)) != len(cols) or any(c == "" or c != c.strip() for c in cols + rows):
        return pd.read_csv(path, index_col=0)          # pandas mangles duplicate / blank names: let it
    if any(r in _PANDAS_NA for r in rows):
        return pd.read_csv(path, index_col=0)          # pandas turns these row labels into NaN: its result, not ours
    if all(_INT.match(r) for r in rows):
        try:
            index = pd.Index(np.array([int(r) for r in rows], np.int64))  # pandas infers an integer index
        except OverflowError:
            return pd.read_csv(path, index_col=0)      # beyond int64: pandas' own inference (uint64 / object)
    else:
        try:
            [float(r) for r in rows]
            return pd.read_csv(path, index_col=0)      # a float-like index: pandas' inference applies
        except ValueError:
            index = pd.Index(rows, dtype=object)
    index.name = name or None
    return pd.DataFrame(values, index=index, columns=pd.Index(cols, dtype=object), copy=False)


def _plain(labels):
    out = []
    for x in labels:
        s = x if isinstance(x, str) else repr(x) if isinstance(x, float) else str(x)
        if any(ch in s for ch in ',"\r\n\0') or s == "":
            return None
        out.append(s)
    return out


def to_csv(frame, path):
    """frame.to_csv(path) for a float64 frame with plain labels; anything else goes to pandas."""
    try:
        fns = _fns()
    except (ImportError, OSError):
        return frame.to_csv(path)
    cols, rows = _plain(frame.columns), _plain(frame.index)
    single = not isinstance(frame.columns, pd.MultiIndex) and not isinstance(frame.index, pd.MultiIndex)
    if cols is None or rows is None or not single or not all(dt == np.float64 for dt in frame.dtypes) or frame.shape[1] == 0:
        return frame.to_csv(path)
    name = frame.index.name
    if name is not None and _plain([name]) is None:
        return frame.to_csv(path)
    values = np.ascontiguousarray(frame.values, dtype=np.float64)
    cb = ("\0".join(cols) + "\0").encode("utf-8")
    rb = ("\0".join(rows) + "\0").encode("utf-8") if rows else b"\0"
    rc = fns["csv_write"](str(path).encode(), values.ctypes.data_as(C.POINTER(C.c_double)), values.shape[0], values.shape[1],
                          ("" if name is None else str(name)).encode("utf-8"), cb, rb)
    if rc != 0:
        raise OSError("dimn_csv_write: " + fns["last_error"]().decode("utf-8", "replace"))
