"""The CSV edges of the CLI on libdimn's host reader / writer (reference deepImpute.py:13, :35) against
pandas itself: the same frame from read_csv, the same BYTES from to_csv.  Host-only: runs without a GPU."""
import numpy as np
import pandas as pd
import pytest

from deepimpute_amd import csvio


def _counts(n, g, seed=0):
    rng = np.random.default_rng(seed)
    return rng.poisson(rng.gamma(0.6, 4.0, size=g), size=(n, g)).astype(np.int64)


@pytest.mark.parametrize("index_kind,eol,tail", [("str", "\n", True), ("int", "\n", False), ("str", "\r\n", True), ("mixed", "\n", True)])
def test_read_csv_matches_pandas(tmp_path, index_kind, eol, tail):
    n, g = 57, 43
    vals = _counts(n, g)
    vals[3, 5] = -7                                                     # a sign
    rows = {"str": ["cell_%d" % i for i in range(n)], "int": [str(1000 - 3 * i) for i in range(n)],
            "mixed": [("%d" % i if i % 2 else "c%d" % i) for i in range(n)]}[index_kind]
    cols = ["GENE%d.1" % j for j in range(g)]
    text = eol.join(["," + ",".join(cols)] + [r + "," + ",".join(str(v) for v in vals[i]) for i, r in enumerate(rows)]) + (eol if tail else "")
    path = tmp_path / "in.csv"
    path.write_bytes(text.encode())
    got, ref = csvio.read_csv(str(path)), pd.read_csv(str(path), index_col=0)
    pd.testing.assert_frame_equal(got, ref, check_exact=True)
    assert got.values.dtype == np.int64 and got.index.dtype == ref.index.dtype and list(got.columns) == cols


def test_read_csv_falls_back_to_pandas_for_anything_else(tmp_path):
    cases = {"decimal": ",a,b\nx,1.5,2\ny,3,4\n", "quoted": ',a,b\n"x,1",1,2\ny,3,4\n', "empty": ",a,b\nx,,2\ny,3,4\n",
             "ragged": ",a,b\nx,1,2\ny,3\n", "named_index": "cell,a,b\nx,1,2\ny,3,4\n", "dup_cols": ",a,a\nx,1,2\ny,3,4\n",
             "float_index": ",a,b\n0.5,1,2\n1.5,3,4\n", "exponent": ",a,b\nx,1e3,2\ny,3,4\n"}
    for name, text in cases.items():
        path = tmp_path / (name + ".csv")
        path.write_text(text)
        pd.testing.assert_frame_equal(csvio.read_csv(str(path)), pd.read_csv(str(path), index_col=0), check_exact=True, obj=name)


def test_to_csv_is_byte_identical_to_pandas(tmp_path):
    rng = np.random.default_rng(1)
    n, g = 211, 37
    mag = 10.0 ** rng.uniform(-9, 21, size=(n, g))
    vals = np.where(rng.random((n, g)) < 0.5, mag, np.rint(rng.uniform(0, 5e4, size=(n, g)))) * rng.choice([1.0, -1.0], size=(n, g))
    vals[0, :12] = [0.0, -0.0, 1e16, 9999999999999998.0, 1e15, 1e-4, 9.999e-5, 1e-5, 123456789012345678.0, 0.1 + 0.2, 1 / 3, 5e-324]
    vals[1, :4] = [np.nan, np.inf, -np.inf, 2.5]
    vals[2] = np.expm1(rng.uniform(0, 12, size=g))                      # what predict() writes
    for index in (pd.Index(["c%d" % i for i in range(n)]), pd.Index(np.arange(n, dtype=np.int64) * 7), pd.Index(["c%d" % i for i in range(n)], name="cell")):
        frame = pd.DataFrame(vals, index=index, columns=["g%d" % j for j in range(g)])
        a, b = tmp_path / "native.csv", tmp_path / "pandas.csv"
        csvio.to_csv(frame, str(a))
        frame.to_csv(str(b))
        assert a.read_bytes() == b.read_bytes()


def test_to_csv_hands_unusual_frames_to_pandas(tmp_path):
    frames = [pd.DataFrame(np.ones((3, 2)), index=['a,b', 'c', 'd'], columns=["x", "y"]),            # a label that needs quoting
              pd.DataFrame(np.ones((3, 2), np.float32), index=list("abc"), columns=["x", "y"]),      # not float64
              pd.DataFrame({"x": [1.0, 2.0], "y": [1, 2]}, index=["a", "b"])]                       # mixed dtypes
    for i, frame in enumerate(frames):
        a, b = tmp_path / ("n%d.csv" % i), tmp_path / ("p%d.csv" % i)
        csvio.to_csv(frame, str(a))
        frame.to_csv(str(b))
        assert a.read_bytes() == b.read_bytes()


def test_round_trip_of_a_larger_matrix_is_threaded_and_exact(tmp_path):
    n, g = 1500, 900                                                    # > 1M fields: the multi-threaded branches
    vals = _counts(n, g, seed=3)
    frame = pd.DataFrame(vals.astype(np.float64) * 1.25, index=["c%d" % i for i in range(n)], columns=["g%d" % j for j in range(g)])
    a, b = tmp_path / "native.csv", tmp_path / "pandas.csv"
    csvio.to_csv(frame, str(a))
    frame.to_csv(str(b))
    assert a.read_bytes() == b.read_bytes()
    ints = pd.DataFrame(vals, index=frame.index, columns=frame.columns)
    ints.to_csv(str(b))
    pd.testing.assert_frame_equal(csvio.read_csv(str(b)), pd.read_csv(str(b), index_col=0), check_exact=True)


def test_row_labels_pandas_treats_as_missing_or_huge_go_to_pandas(tmp_path):
    """Row labels in pandas' NA set become NaN in pd.read_csv's index, all-integer labels beyond int64 become uint64/object:
    in both cases read_csv must return exactly what pandas returns."""
    import pandas as pd
    from deepimpute_amd import csvio
    for name, rows in (("na", ["c0", "NA", "null", "c3"]), ("big", ["1", "99999999999999999999999", "3", "4"])):
        path = str(tmp_path / (name + ".csv"))
        with open(path, "w") as f:
            f.write(",g0,g1\n")
            for i, r in enumerate(rows):
                f.write("%s,%d,%d\n" % (r, i, 2 * i))
        want = pd.read_csv(path, index_col=0)
        got = csvio.read_csv(path)
        pd.testing.assert_frame_equal(got, want)


def test_reader_line_table_handles_blank_lines_and_chunk_boundaries(tmp_path):
    """The reader finds its lines with memchr on several threads (round 5): blank lines anywhere (pandas skips them), "\\r\\n", no final
    newline, and a file big enough (> 8 MB) that line boundaries fall inside and across the threads' chunks -- frame equal to pandas'."""
    rng = np.random.default_rng(7)
    n, g = 3000, 1500
    a = rng.poisson(4.0, size=(n, g))
    rows = ["c%d," % i + ",".join(map(str, a[i])) for i in range(n)]
    head = "," + ",".join("g%d" % j for j in range(g))
    for name, text in (("big_lf", "\n".join([head] + rows) + "\n"),
                       ("blank", "\r\n".join([head, ""] + rows[:50] + ["", ""] + rows[50:200]) + "\r\n\r\n"),
                       ("no_final_newline", "\n".join([head] + rows[:10]))):
        path = tmp_path / (name + ".csv")
        path.write_text(text, newline="")
        if name == "big_lf":
            assert path.stat().st_size > (8 << 20)
        ours, theirs = csvio.read_csv(str(path)), pd.read_csv(str(path), index_col=0)
        pd.testing.assert_frame_equal(ours, theirs)
        assert ours.values.dtype == np.int64
