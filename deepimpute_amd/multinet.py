"""Drop-in `MultiNet` estimator on the MI355X-native engine.

Same public surface as the reference's `deepimpute.multinet` (class MultiNet with
fit/predict/score, module functions get_distance_matrix / wMSE / inspect_data; reference
deepimpute/multinet.py:20-63, 65-374).  Everything the reference hands to Keras/TensorFlow
(build -> compile -> model.fit -> model.predict, multinet.py:126-167, 238-253, 276-280) goes
to `deepimpute_amd.engine.HipEngine` instead, i.e. to hand-written gfx950 kernels behind the C
ABI of include/dimn.h.  The host-side gene selection / predictor selection / post-processing
around that seam is restated here with numpy/pandas so that the global-numpy-RNG call order,
the printed messages and the returned frames match the reference (pinned by
tests/golden/shell_*.npz, captured from the imported reference).

There is no CPU fallback: without libdimn.so + a GPU, fit()/predict() raise.
"""
import json
import os
import tempfile
import warnings

import numpy as np
import pandas as pd
from scipy.stats import pearsonr

_DEFAULT_OUTPUT_PREFIX = tempfile.mkdtemp()   # one temp dir per import, as in the reference (:74)


def get_distance_matrix(raw, n_pred=None):
    """|Pearson| between genes with non-zero variance/mean ratio (reference multinet.py:20-34)."""
    vmr = raw.std() / raw.mean()
    vmr[np.isinf(vmr)] = 0
    if n_pred is None:
        candidates = raw.columns[vmr > 0]
    else:
        print("Using {} predictors".format(n_pred))
        candidates = vmr.sort_values(ascending=False).index[:n_pred]
    corr = np.abs(np.corrcoef(raw.T.loc[candidates]))
    return pd.DataFrame(corr, index=candidates, columns=candidates).fillna(0)


def wMSE(y_true, y_pred, binary=False):
    """Weighted MSE of the reference (multinet.py:36-41) as a numpy function: mean over all
    elements of w*(y-yhat)^2 with w = y_true (or 1[y_true>0]).  The training kernels implement
    exactly this; the function is kept for API compatibility and for checks."""
    y_true = np.asarray(y_true)
    y_pred = np.asarray(y_pred)
    weights = (y_true > 0).astype(np.float32) if binary else y_true
    return np.mean(weights * np.square(y_true - y_pred))


def inspect_data(data):
    """Input guards of the reference (multinet.py:43-63): unique labels, raw counts."""
    if sum(data.index.duplicated()):
        print("ERROR: duplicated cell labels. Please provide unique cell labels.")
        exit(1)
    if sum(data.columns.duplicated()):
        print("ERROR: duplicated gene labels. Please provide unique gene labels.")
        exit(1)
    max_value = np.max(data.values)
    if max_value < 10:
        print("ERROR: max value = {}. Is your data log-transformed? Please provide raw counts"
              .format(max_value))
        exit(1)
    print("Input dataset is {} cells (rows) and {} genes (columns)".format(*data.shape))
    print("First 3 rows and columns:")
    print(data.iloc[:3, :3])


_LOSSES = {"wMSE": 0, "wmse": 0}


def _parse_architecture(architecture):
    """The kernels implement Dense(H, relu) [-> Dropout(p)] -> Dense(O, softplus), the only form
    any caller of the reference uses (multinet.py:99-103, deepImpute.py:24-26).  Returns (H, p)."""
    hidden, rate, seen_dropout = None, 0.0, False
    for layer in architecture:
        kind = str(layer.get("type", "")).lower()
        if kind == "dense":
            if hidden is not None or seen_dropout:
                raise NotImplementedError(
                    "deepimpute_amd supports one hidden dense layer followed by an optional dropout; "
                    "got architecture %r" % (architecture,))
            if str(layer.get("activation", "relu")).lower() != "relu":
                raise NotImplementedError("hidden activation %r is not implemented (relu only)"
                                          % (layer.get("activation"),))
            hidden = int(layer["neurons"])
        elif kind == "dropout":
            if hidden is None or seen_dropout:
                raise NotImplementedError("dropout must follow the hidden dense layer, once")
            rate = float(layer["rate"])
            seen_dropout = True
        else:
            print("Unknown layer type.")   # reference multinet.py:142-143 ignores it
    if hidden is None:
        raise NotImplementedError("architecture needs one hidden dense layer")
    return hidden, rate


class MultiNet:
    def __init__(self,
                 learning_rate=1e-4,
                 batch_size=64,
                 max_epochs=500,
                 patience=5,
                 ncores=-1,
                 loss="wMSE",
                 output_prefix=_DEFAULT_OUTPUT_PREFIX,
                 sub_outputdim=512,
                 verbose=1,
                 seed=1234,
                 architecture=None,
                 device_id=0,
                 engine_factory=None):
        # same hyper-parameter dict as the reference (multinet.py:80-85)
        self.NN_parameters = {"learning_rate": learning_rate,
                              "batch_size": batch_size,
                              "loss": loss,
                              "architecture": architecture,
                              "max_epochs": max_epochs,
                              "patience": patience}
        self.sub_outputdim = sub_outputdim
        self.outputdir = output_prefix
        self.verbose = verbose
        self.seed = seed
        self.device_id = device_id
        self._engine_factory = engine_factory   # test hook; None -> HipEngine (GPU, no fallback)
        self._engine = None
        self.setCores(ncores)

    # ncores only sets TF's CPU thread pools in the reference (multinet.py:222-223); the GPU
    # path has no use for it but the attribute and the message are kept.
    def setCores(self, ncores):
        if ncores > 0:
            self.ncores = ncores
        else:
            self.ncores = os.cpu_count()
            print("Using all the cores ({})".format(self.ncores))

    def loadDefaultArchitecture(self):
        self.NN_parameters['architecture'] = [
            {"type": "dense", "neurons": self.sub_outputdim // 2, "activation": "relu"},
            {"type": "dropout", "rate": 0.2},
        ]

    # ---- engine construction: stands where build() creates the Keras model (:126-167) ----
    def build(self, inputdims, subnet_offset=0):
        if self.NN_parameters['architecture'] is None:
            self.loadDefaultArchitecture()
        print(self.NN_parameters['architecture'])
        hidden, rate = _parse_architecture(self.NN_parameters['architecture'])
        loss = self.NN_parameters['loss']
        if callable(loss):
            loss = getattr(loss, "__name__", "")
        if loss not in _LOSSES:
            print('Unknown loss: {}. Aborting.'.format(loss))
            exit(1)
        factory = self._engine_factory
        if factory is None:
            from .engine import HipEngine
            factory = HipEngine
        return factory(list(inputdims), hidden, self.sub_outputdim,
                       batch_size=self.NN_parameters["batch_size"], dropout_rate=rate,
                       learning_rate=self.NN_parameters["learning_rate"],
                       seed=self.seed if self.seed is not None else 0,
                       device_id=self.device_id, subnet_offset=subnet_offset)

    # ---- persistence: model.json + weights (reference writes model.json + model.h5) ----
    def save(self, model):
        os.makedirs(self.outputdir, exist_ok=True)
        hidden, rate = _parse_architecture(self.NN_parameters['architecture'])
        meta = {"format": "deepimpute_amd-1", "inputdims": list(model.D), "hidden": hidden,
                "dropout_rate": rate, "sub_outputdim": self.sub_outputdim,
                "architecture": self.NN_parameters['architecture']}
        with open("{}/model.json".format(self.outputdir), "w") as json_file:
            json.dump(meta, json_file)
        arrays = {}
        for k in range(model.K):
            W1, b1, W2, b2 = model.get_weights(k)
            arrays["W1_%d" % k], arrays["b1_%d" % k] = W1, b1
            arrays["W2_%d" % k], arrays["b2_%d" % k] = W2, b2
        np.savez("{}/model.npz".format(self.outputdir), **arrays)
        print("Saved model to disk in {}".format(self.outputdir))

    def load(self):
        """Engine with the weights saved by fit() (reference load(): model_from_json + load_weights,
        multinet.py:117-124).  Re-uses the live engine when this object trained it."""
        if self._engine is not None:
            return self._engine
        with open('{}/model.json'.format(self.outputdir), 'r') as json_file:
            meta = json.load(json_file)
        self.NN_parameters['architecture'] = meta["architecture"]
        self.sub_outputdim = meta["sub_outputdim"]
        model = self.build(meta["inputdims"])
        with np.load('{}/model.npz'.format(self.outputdir)) as z:
            for k in range(model.K):
                model.set_weights(k, z["W1_%d" % k], z["b1_%d" % k], z["W2_%d" % k], z["b2_%d" % k])
        self._engine = model
        return model

    def _set_columns(self, model, columns):
        col_index = pd.Index(columns)
        for k in range(model.K):
            p = col_index.get_indexer(self.predictors[k])
            t = col_index.get_indexer(self.targets[k])
            if (p < 0).any() or (t < 0).any():
                raise KeyError("predictor/target genes missing from the data columns")
            model.set_indices(k, p, t)

    def fit(self,
            raw,
            cell_subset=1,
            NN_lim=None,
            genes_to_impute=None,
            n_pred=None,
            ntop=5,
            minVMR=0.5,
            mode='random'):
        inspect_data(raw)

        if self.seed is not None:
            np.random.seed(self.seed)

        if cell_subset != 1:
            if cell_subset < 1:
                raw = raw.sample(frac=cell_subset)
            else:
                raw = raw.sample(int(cell_subset))   # the CLI passes a float (parser.py:38)

        gene_metric = (raw.var() / (1 + raw.mean())).sort_values(ascending=False)
        gene_metric = gene_metric[gene_metric > 0]

        if genes_to_impute is None:
            genes_to_impute = self.filter_genes(gene_metric, minVMR, NN_lim=NN_lim)
        else:
            n_genes = len(genes_to_impute)
            if n_genes % self.sub_outputdim != 0:
                print("The number of input genes is not a multiple of {}. Filling with other genes.".format(n_genes))
                fill_genes = gene_metric.index[:self.sub_outputdim - n_genes]
                if len(fill_genes) < self.sub_outputdim - n_genes:
                    rest = self.sub_outputdim - n_genes - len(fill_genes)
                    fill_genes = np.concatenate([fill_genes,
                                                 np.random.choice(gene_metric.index, rest, replace=True)])
                genes_to_impute = np.concatenate([genes_to_impute, fill_genes])

        covariance_matrix = get_distance_matrix(raw, n_pred=n_pred)

        self.setTargets(raw.reindex(columns=genes_to_impute), mode=mode)
        self.setPredictors(covariance_matrix, ntop=ntop)

        print("Normalization")
        norm_data = np.log1p(raw).astype(np.float32)

        np.random.seed(self.seed)

        print("Building network")
        if self._engine is not None:
            self._engine.close()
            self._engine = None
        model = self.build([len(genes) for genes in self.predictors])

        test_cells = np.random.choice(norm_data.index, int(0.05 * norm_data.shape[0]), replace=False)
        train_cells = np.setdiff1d(norm_data.index, test_cells)

        # the reference materialises 4K host arrays here (multinet.py:231-235); the engine
        # takes the shared matrix once plus index lists and gathers on the device
        model.set_matrix(norm_data.values)
        self._set_columns(model, norm_data.columns)
        model.gather(True)
        train_rows = norm_data.index.get_indexer(train_cells)
        test_rows = norm_data.index.get_indexer(test_cells)
        model.set_split(train_rows, test_rows)
        model.init_weights(self.seed if self.seed is not None else 0)

        print("Fitting with {} cells".format(norm_data.shape[0]))
        epochs_run, loss_hist, val_hist = model.fit(self.NN_parameters["max_epochs"],
                                                    self.NN_parameters["patience"])
        self.history = {"loss": list(loss_hist), "val_loss": list(val_hist)}
        if self.verbose:
            for e, (l, v) in enumerate(zip(loss_hist, val_hist)):
                print("Epoch {}/{} - loss: {:.4f} - val_loss: {:.4f}".format(
                    e + 1, self.NN_parameters["max_epochs"], l, v))

        self.trained_epochs = epochs_run
        print("Stopped fitting after {} epochs".format(self.trained_epochs))

        self._engine = model
        self.save(model)

        # held-out metrics on the validation cells (reference multinet.py:251-262)
        Y_test_raw = np.hstack([norm_data.loc[test_cells, t].values for t in self.targets]).flatten()
        Y_test_imputed = model.predict(test_rows).flatten()
        Y_test_imputed = Y_test_imputed[Y_test_raw > 0]
        Y_test_raw = Y_test_raw[Y_test_raw > 0]
        self.test_metrics = {
            'correlation': pearsonr(Y_test_raw, Y_test_imputed)[0],
            'MSE': np.sum((Y_test_raw - Y_test_imputed) ** 2) / len(Y_test_raw)
        }
        return self

    def predict(self,
                raw,
                imputed_only=False,
                policy="restore"):
        norm_raw = np.log1p(raw)

        model = self.load()
        model.set_matrix(norm_raw.values.astype(np.float32))
        self._set_columns(model, norm_raw.columns)
        model.gather(False)
        predicted = model.predict()             # [cells, K*O] == np.hstack(model.predict(inputs))

        # duplicated target genes are averaged; columns come back label-sorted (the
        # reference's groupby(by=columns, axis=1).mean(), multinet.py:282-284)
        flat_targets = self.targets.flatten()
        uniq, inverse = np.unique(flat_targets, return_inverse=True)
        counts = np.bincount(inverse, minlength=len(uniq)).astype(np.float32)
        summed = np.zeros((predicted.shape[0], len(uniq)), dtype=np.float32)
        np.add.at(summed.T, inverse, predicted.T)
        predicted = pd.DataFrame(summed / counts, index=raw.index, columns=uniq)
        not_predicted = norm_raw.drop(uniq, axis=1)

        imputed = (pd.concat([predicted, not_predicted], axis=1)
                   .loc[raw.index, raw.columns]
                   .values)

        # To prevent overflow (multinet.py:292), then back to counts
        imputed[(imputed > 2 * norm_raw.values.max()) | (np.isnan(imputed))] = 0
        imputed = np.expm1(imputed)

        if policy == "restore":
            print("Filling zeros")
            mask = (raw.values > 0)
            imputed[mask] = raw.values[mask]
        elif policy == "max":
            print("Imputing data with 'max' policy")
            mask = (raw.values > imputed)
            imputed[mask] = raw.values[mask]

        imputed = pd.DataFrame(imputed, index=raw.index, columns=raw.columns)

        if imputed_only:
            return imputed.loc[:, predicted.columns]
        else:
            return imputed

    def filter_genes(self,
                     gene_metric,   # assumes gene_metric is sorted
                     threshold,
                     NN_lim=None):
        if not str(NN_lim).isdigit():
            NN_lim = (gene_metric > threshold).sum()
        NN_lim = int(NN_lim)           # the CLI hands a digit string (parser.py:26)

        n_subsets = int(np.ceil(NN_lim / self.sub_outputdim))
        genes_to_impute = gene_metric.index[:n_subsets * self.sub_outputdim]

        rest = self.sub_outputdim - (len(genes_to_impute) % self.sub_outputdim)
        if rest > 0:
            fill_genes = np.random.choice(gene_metric.index, rest)
            genes_to_impute = np.concatenate([genes_to_impute, fill_genes])

        print("{} genes selected for imputation".format(len(genes_to_impute)))
        return genes_to_impute

    def setTargets(self, data, mode='random'):
        n_subsets = int(data.shape[1] / self.sub_outputdim)
        if mode == 'progressive':
            self.targets = data.columns.values.reshape([n_subsets, self.sub_outputdim])
        else:
            self.targets = np.random.choice(data.columns,
                                            [n_subsets, self.sub_outputdim],
                                            replace=False)

    def setPredictors(self, covariance_matrix, ntop=5):
        """Top-`ntop` most correlated non-target genes per target, first-occurrence order
        (reference multinet.py:344-365).  The reference argsorts every full row; a partial
        selection gives the same genes in the same order whenever correlations are distinct."""
        self.predictors = []
        all_cols = covariance_matrix.columns
        for i, targets in enumerate(self.targets):
            genes_not_in_target = np.setdiff1d(all_cols, targets)
            if genes_not_in_target.size == 0:
                warnings.warn('Warning: number of target genes lower than output dim. '
                              'Consider lowering down the sub_outputdim parameter', UserWarning)
                genes_not_in_target = all_cols
            sub = covariance_matrix.loc[targets, genes_not_in_target].values
            take = min(ntop, sub.shape[1])
            if take < sub.shape[1]:
                part = np.argpartition(-sub, take - 1, axis=1)[:, :take]
            else:
                part = np.tile(np.arange(sub.shape[1]), (sub.shape[0], 1))
            vals = np.take_along_axis(sub, part, axis=1)
            order = np.argsort(-vals, axis=1, kind="stable")
            top = np.take_along_axis(part, order, axis=1)
            predictors = pd.Index(genes_not_in_target)[top.flatten()]
            self.predictors.append(predictors.unique())
            print("Net {}: {} predictors, {} targets".format(i, len(np.unique(predictors)), len(targets)))

    def score(self, data, policy=None):
        warnings.warn(
            "This method is deprecated. Please use model.test_metrics to measure model accuracy instead",
            DeprecationWarning)
        Y_hat = self.predict(data, policy=policy)
        Y = data.loc[Y_hat.index, Y_hat.columns]
        return pearsonr(Y_hat.values.reshape(-1), Y.values.reshape(-1))
