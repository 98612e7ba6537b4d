"""`deepImpute` entry point (function + console script), same call shape as the reference's
deepimpute/deepImpute.py:6-40: parse flags, let keyword arguments override them, read the CSV,
fit a MultiNet on the GPU, impute, write or return the result."""
import os
import sys
import time

_T_IMPORT = time.time()                 # (DIMN_TRACE: when this module began to import its dependencies)

if __name__ == "__main__" and not {"-h", "--help"} & set(sys.argv[1:]):
    # `python -m deepimpute_amd.deepImpute ...`: the process exists to use the GPU, so the HIP runtime loads and the device comes up
    # (dimn_warm_up: context, pinned bounce buffers, the epilogue's blocks) on a helper thread WHILE pandas and numpy are still importing
    # -- on a cold box each takes seconds.  Importing this module from other code starts nothing.
    try:
        from . import _lib as _early_lib
        _early_lib.warm_up_async(0)
    except (ImportError, OSError):
        pass                            # no library / no GPU: MultiNet.fit says so

from . import csvio                     # noqa: E402
from .multinet import MultiNet          # noqa: E402
from .parser import parse_args          # noqa: E402


def _trace(marks, net):
    """DIMN_TRACE=1: wall-clock marks of the run on stderr (absolute times, so that a caller can add the interpreter's own start-up),
    followed by the stage times of fit() / predict() -- tools/cli_cold.sh reads them."""
    out = ["[deepImpute] t0 %.6f" % marks[0][1]]
    out += ["[deepImpute] %-28s %8.3f s" % (name, t - marks[i][1]) for i, (name, t) in enumerate(marks[1:])]
    out += ["[deepImpute]   %-26s %8.3f s" % kv for kv in sorted(getattr(net, "timings", {}).items())]
    sys.stderr.write("\n".join(out) + "\n")


def deepImpute(**kwargs):
    marks = [("process imports", _T_IMPORT), ("imports (pandas, numpy, scipy)", time.time())]
    args = parse_args()                 # always parses sys.argv, as the reference does
    for name, value in kwargs.items():
        setattr(args, name, value)

    try:                                                # the GPU comes up (HIP context, pinned buffers) while the CSV is parsed
        from . import _lib
        _lib.warm_up_async(0)
    except (ImportError, OSError):
        pass                                            # no library / no GPU: MultiNet.fit says so
    counts = csvio.read_csv(args.inputFile)             # pd.read_csv(inputFile, index_col=0), multi-threaded for count matrices
    if args.cell_axis == "columns":
        counts = counts.T
    marks.append(("read_csv", time.time()))

    net = MultiNet(learning_rate=args.learning_rate,
                   batch_size=args.batch_size,
                   max_epochs=args.max_epochs,
                   ncores=args.cores,
                   sub_outputdim=args.output_neurons,
                   architecture=[{"type": "dense", "activation": "relu", "neurons": args.hidden_neurons},
                                 {"type": "dropout", "activation": "dropout", "rate": args.dropout_rate}])
    net.fit(counts, NN_lim=args.limit, cell_subset=args.subset, minVMR=args.minVMR, n_pred=args.n_pred)
    marks.append(("fit", time.time()))
    fit_stages = dict(getattr(net, "timings", {}))
    imputed = net.predict(counts, imputed_only=False, policy=args.policy)
    marks.append(("predict", time.time()))

    if args.output is not None:
        csvio.to_csv(imputed, args.output)              # imputed.to_csv(output), multi-threaded
        marks.append(("to_csv", time.time()))
    if os.environ.get("DIMN_TRACE", "0") not in ("", "0"):
        net.timings = dict(fit_stages, **getattr(net, "timings", {}))
        _trace(marks, net)
    if args.output is None:
        return imputed


if __name__ == "__main__":
    deepImpute()
