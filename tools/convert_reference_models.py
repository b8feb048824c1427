"""Convert model folders written by the reference package (two pickles per label:
log_trans_probs.pickle, multivariate_normals.pickle; hidden_markov_model.py:93-115 of the
reference) into the flat `<label>/model.npz` files this package reads.

The pickles hold instances of the reference's own classes wrapping SciPy frozen
distributions, so unpickling needs the reference sources importable.  Only use it on model
folders you produced yourself: unpickling executes code from the file.

    python tools/convert_reference_models.py --reference /path/to/CS-304-Speech-Recognition-Code \
        --src .cache/big_model --dst converted/big_model
"""
import argparse
import importlib
import os
import pickle
import sys
import types
import typing

import numpy as np


def import_reference(root: str):
    """Load the reference's HMM modules without its package __init__ (which needs librosa,
    sounddevice and Python >= 3.12)."""
    import typing_extensions

    if not hasattr(typing, "Self"):
        typing.Self = typing_extensions.Self
    sys.modules.setdefault("uniplot", types.ModuleType("uniplot"))
    pkg = types.ModuleType("loe_speech_recognition")
    pkg.__path__ = [os.path.join(root, "src", "loe_speech_recognition")]
    sys.modules["loe_speech_recognition"] = pkg
    return importlib.import_module("loe_speech_recognition.hidden_markov_model")


def convert_folder(src: str, dst: str) -> str:
    label = src.rstrip("/").split("/")[-1]
    with open(os.path.join(src, "log_trans_probs.pickle"), "rb") as f:
        ltp = pickle.load(f)
    with open(os.path.join(src, "multivariate_normals.pickle"), "rb") as f:
        mvns = pickle.load(f)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", required=True, help="checkout of the reference repository")
    ap.add_argument("--src", required=True, help="folder with one sub-folder per label (the reference's save() layout)")
    ap.add_argument("--dst", required=True)
    args = ap.parse_args()
    sys.dont_write_bytecode = True
    import_reference(args.reference)
    for name in sorted(os.listdir(args.src)):
        path = os.path.join(args.src, name)
        if os.path.isfile(os.path.join(path, "multivariate_normals.pickle")):
            print("wrote", convert_folder(path, args.dst))


if __name__ == "__main__":
    main()
