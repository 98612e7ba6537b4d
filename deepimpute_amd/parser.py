"""Command-line flags of `deepImpute` (same 14 options, names, types and defaults as the
reference's deepimpute/parser.py:3-95, including its defaults that differ from the Python API:
learning rate 5e-4, 300 epochs, 300 hidden units)."""
import argparse

# (flags, keyword arguments) -- kept as a table so the CLI surface is readable at a glance
_OPTIONS = [
    (("inputFile",), dict(type=str, help="Path to input data.")),
    (("-o", "--output"), dict(type=str, default="./imputed.csv",
                              help="Path to output data counts. Default: ./imputed.csv")),
    (("--cores",), dict(type=int, default=-1, help="Number of cores. Default: all available cores")),
    (("--cell-axis",), dict(type=str, choices=["rows", "columns"], default="rows",
                            help="Cell dimension in the matrix. Default: rows")),
    (("--limit",), dict(type=str, default="auto",
                        help="Genes to impute (e.g. first 2000 genes). Default: auto")),
    (("--minVMR",), dict(type=float, default="0.5",
                         help="Min Variance over mean ratio for gene exclusion. Gene with a VMR below "
                              "${minVMR} are discarded. Used if --limit is set to 'auto'. Default: 0.5")),
    (("--subset",), dict(type=float, default=1,
                         help="Cell subset to speed up training. Either a ratio (0<x<1) or a cell "
                              "number (int). Default: 1 (all)")),
    (("--learning-rate",), dict(type=float, default=0.0005, help="Learning rate. Default: 0.0001")),
    (("--batch-size",), dict(type=int, default=64, help="Batch size. Default: 64")),
    (("--max-epochs",), dict(type=int, default=300, help="Maximum number of epochs. Default: 500")),
    (("--hidden-neurons",), dict(type=int, default=300,
                                 help="Number of neurons in the hidden dense layer. Default: 256")),
    (("--dropout-rate",), dict(type=float, default=0.2,
                               help="Dropout rate for the hidden dropout layer (0<rate<1). Default: 0.2")),
    (("--output-neurons",), dict(type=int, default=512,
                                 help="Number of output neurons per sub-network. Default: 512")),
    (("--n_pred",), dict(type=int, default=None,
                         help="Number of predictors to consider. Consider using this parameter if your "
                              "RAM is limited or if you have a high number of features. Default: All "
                              "genes with nonzero VMR")),
    (("--policy",), dict(type=str, default='restore',
                         help="Whether to restore positive values from the raw dataset or keep the max "
                              "between the imputed values and the raw values. Choices are "
                              "['restore', 'max']. Default: restore")),
]


def build_parser():
    parser = argparse.ArgumentParser(description="scRNA-seq data imputation using DeepImpute.")
    for flags, kw in _OPTIONS:
        parser.add_argument(*flags, **kw)
    return parser


def parse_args():
    return build_parser().parse_args()
