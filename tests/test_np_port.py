"""Pins the BLAS-backed CPU port used for the timed baseline (oracle/np_port.py) against the
plain-loop oracle and the torch-fp64 known answers: same steps, same masks."""
import numpy as np

from helpers import load_kat, kat_engine, run_kat_steps
from oracle.dimo import OracleEngine
from oracle.np_port import NumpyPort


def _adapt(p):
    p.set_split = lambda tr, va: None
    p.reset_optimizer = lambda: None
    return p


def test_np_port_matches_oracle_and_kat():
    kat = load_kat()
    port = kat_engine(lambda D, H, O, **kw: _adapt(NumpyPort(D, H, O, **kw)), kat)
    ora = kat_engine(OracleEngine, kat)
    la, lb = run_kat_steps(port, kat), run_kat_steps(ora, kat)
    np.testing.assert_allclose(la, lb, rtol=1e-4)
    for k in range(len(kat["Ds"])):
        np.testing.assert_allclose(la[:, k], kat["loss_%d" % k], rtol=2e-4)
        for a, b, name in zip(port.get_weights(k), ora.get_weights(k), ("W1", "b1", "W2", "b2")):
            np.testing.assert_allclose(a, b, rtol=2e-3, atol=2e-6, err_msg=name)
            np.testing.assert_allclose(a, kat["out_%s_%d" % (name, k)], rtol=2e-3, atol=2e-6, err_msg=name)
    np.testing.assert_allclose(port.predict(), ora.predict(), rtol=2e-4, atol=1e-6)
