"""`deepImpute` entry point (function + console script), same call shape as the reference's
deepimpute/deepImpute.py:6-40: parse flags, let keyword arguments override them, read the CSV,
fit a MultiNet on the GPU, impute, write or return the result."""
from . import csvio
from .multinet import MultiNet
from .parser import parse_args


def deepImpute(**kwargs):
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

    net = MultiNet(learning_rate=args.learning_rate,
                   batch_size=args.batch_size,
                   max_epochs=args.max_epochs,
                   ncores=args.cores,
                   sub_outputdim=args.output_neurons,
                   architecture=[{"type": "dense", "activation": "relu", "neurons": args.hidden_neurons},
                                 {"type": "dropout", "activation": "dropout", "rate": args.dropout_rate}])
    net.fit(counts, NN_lim=args.limit, cell_subset=args.subset, minVMR=args.minVMR, n_pred=args.n_pred)
    imputed = net.predict(counts, imputed_only=False, policy=args.policy)

    if args.output is None:
        return imputed
    csvio.to_csv(imputed, args.output)                  # imputed.to_csv(output), multi-threaded


if __name__ == "__main__":
    deepImpute()
