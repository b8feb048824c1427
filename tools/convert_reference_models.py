"""Convert model folders written by the reference package (two pickles per label:
log_trans_probs.pickle, multivariate_normals.pickle; hidden_markov_model.py:93-115 of the
reference) into the flat `<label>/model.npz` files this package reads.

The pickles hold instances of the reference's own classes wrapping SciPy frozen distributions.
They are read with a restricted unpickler: the two reference classes are replaced by local
attribute bags, and the only other globals a file may name are the NumPy array / scalar / random-state
reconstructors and the installed SciPy classes a frozen multivariate normal is made of.  Anything
else (`os.system`, `builtins.eval`, ...) raises UnpicklingError before it is looked up, so the
reference sources are neither imported nor executed.

    python tools/convert_reference_models.py --src .cache/big_model --dst converted/big_model
"""
import argparse
import importlib
import io
import os
import pickle

import numpy as np


class _Bag:
    """Stand-in for a reference dataclass: unpickling only fills its __dict__."""


class LogTransitionProbabilities(_Bag):
    pass


class MultivariateNormal(_Bag):
    pass


_STUBS = {
    ("loe_speech_recognition.transition_probability", "LogTransitionProbabilities"): LogTransitionProbabilities,
    ("loe_speech_recognition.hidden_markov_model", "MultivariateNormal"): MultivariateNormal,
}
# (module, name) pairs of installed library code that a pickled frozen multivariate normal refers to;
# numpy.core.* are the NumPy 1.x spellings of numpy._core.*
_ALLOWED = {
    ("numpy", "dtype"), ("numpy", "ndarray"),
    ("numpy._core.multiarray", "scalar"), ("numpy._core.multiarray", "_reconstruct"), ("numpy._core.numeric", "_frombuffer"),
    ("numpy.core.multiarray", "scalar"), ("numpy.core.multiarray", "_reconstruct"), ("numpy.core.numeric", "_frombuffer"),
    ("numpy.random._pickle", "__randomstate_ctor"), ("numpy.random._pickle", "__bit_generator_ctor"),
    ("numpy.random._pickle", "__generator_ctor"), ("numpy.random._mt19937", "MT19937"),
    ("numpy.random.bit_generator", "__pyx_unpickle_SeedSequence"), ("numpy.random.bit_generator", "SeedSequence"),
    ("numpy.random.mtrand", "RandomState"),
    ("scipy.stats._multivariate", "multivariate_normal_frozen"), ("scipy.stats._multivariate", "multivariate_normal_gen"),
    ("scipy.stats._multivariate", "_PSD"), ("scipy.stats._covariance", "CovViaPSD"),
    ("scipy.stats._covariance", "CovViaEigendecomposition"), ("scipy.stats._covariance", "CovViaCholesky"),
    ("scipy.stats._covariance", "CovViaDiagonal"), ("scipy.stats._covariance", "CovViaPrecision"),
}


class RestrictedUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        key = (module, name)
        if key in _STUBS:
            return _STUBS[key]
        if key in _ALLOWED:
            return getattr(importlib.import_module(module), name)
        raise pickle.UnpicklingError(f"model file refers to {module}.{name}, which a saved model never needs; refusing to load it")


def restricted_load(path: str):
    with open(path, "rb") as f:
        return RestrictedUnpickler(io.BytesIO(f.read())).load()


def convert_folder(src: str, dst: str) -> str:
    label = src.rstrip("/").split("/")[-1]
    ltp = restricted_load(os.path.join(src, "log_trans_probs.pickle"))
    mvns = restricted_load(os.path.join(src, "multivariate_normals.pickle"))
    if not isinstance(ltp, LogTransitionProbabilities) or not isinstance(mvns, list) or \
            not all(isinstance(m, MultivariateNormal) for m in mvns):
        raise ValueError(f"{src}: not a model folder written by the reference's save()")
    n = len(mvns)
    mean = np.stack([np.asarray(m._core.mean, dtype=np.float64) for m in mvns])
    whiten = np.stack([np.asarray(m._core.cov_object._LP, dtype=np.float64) for m in mvns])
    log_pdet = np.array([float(m._core.cov_object._log_pdet) for m in mvns])
    dense = np.full((n, n), np.nan, dtype=np.float32)
    for (i, j), v in ltp._core.items():
        dense[i, j] = v
    os.makedirs(os.path.join(dst, label), exist_ok=True)
    out = os.path.join(dst, label, "model.npz")
    np.savez(out, label=np.array(label), mean=mean, whiten=whiten, log_pdet=log_pdet, log_trans=dense)
    return out


def convert_tree(src: str, dst: str):
    done = []
    for name in sorted(os.listdir(src)):
        path = os.path.join(src, name)
        if os.path.isfile(os.path.join(path, "multivariate_normals.pickle")):
            done.append(convert_folder(path, dst))
    return done


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--src", required=True, help="folder with one sub-folder per label (the reference's save() layout)")
    ap.add_argument("--dst", required=True)
    ap.add_argument("--reference", default=None, help="ignored (kept for older command lines): the reference is not imported")
    args = ap.parse_args()
    for out in convert_tree(args.src, args.dst):
        print("wrote", out)


if __name__ == "__main__":
    main()
