"""Shared helpers for the parity tests: build product-side models from fixture arrays."""
import os

import numpy as np

from loe_speech_recognition_amd import (HiddenMarkovModelInference, HiddenMarkovModelTrainable,
                                        TransitionProbabilities)

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LOOP_LABELS = ["1", "2", "3", "4", "5", "6", "7", "8", "9", "O", "S", "Z"]
BETWEEN = {"m100": -100, "m250": -250, "dflt": np.log(0.005), "m3f": -3.0}


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def trainable(label, means, cov, trans):
    m = HiddenMarkovModelTrainable(label, isTqdm=False)
    m._means = np.array(means, dtype=np.float32)
    m._covariances = np.array(cov, dtype=np.float32)
    m._transition_probs = TransitionProbabilities.from_transition_probability(np.array(trans, dtype=np.float32))
    m._update_inference_weights()
    return m


def word(g, prefix, label=None):
    return trainable(label or prefix, g[f"{prefix}_means"], g[f"{prefix}_cov"], g[f"{prefix}_trans"])


def loop_decoder(g, labels=LOOP_LABELS):
    return HiddenMarkovModelInference.from_models([word(g, lab) for lab in labels])


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))) if a.size else 0.0


def trans_close(a, b, tol=1e-4):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.array_equal(np.isnan(a), np.isnan(b)) and np.allclose(np.nan_to_num(a), np.nan_to_num(b), rtol=tol, atol=tol)
