"""Generate tests/golden/gmm_g9_emission.npz by running the REAL HMM_GMM.log_emission_prob of the reference's
retired mixture code (src/loe_speech_recognition/deprecated/gaussian_mixture_model.py:159-164).

Runs only in the build container, where /root/reference is mounted.  That module cannot be imported the normal
way: it needs librosa, imports `compute_mfcc` from a sibling module that does not exist, and below its class it
runs a training script over ./recordings/...  So: stand-in modules are registered for `librosa` and for the missing
sibling (neither is ever called), the working directory is an empty temporary folder, and the module is executed
until its script part fails on the missing recordings (an ordinary FileNotFoundError).  By then the class HMM_GMM is
defined in the module's namespace; its log_emission_prob method - SciPy's multivariate_normal.logpdf per component,
np.logaddexp over the components, float64 - is called on seeded mixtures.  Nothing from the reference is copied:
the fixture holds the seeded inputs and the values the reference's method returned.

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden_gmm.py
"""
import importlib.util
import os
import sys
import tempfile
import types

import numpy as np

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference/src/loe_speech_recognition"


def load_reference_class():
    def never(*a, **k):
        raise AssertionError("stand-in called")

    librosa = types.ModuleType("librosa")
    librosa.load = never
    sys.modules["librosa"] = librosa
    pkg = types.ModuleType("loe_speech_recognition")
    pkg.__path__ = [SRC]
    sys.modules["loe_speech_recognition"] = pkg
    dep = types.ModuleType("loe_speech_recognition.deprecated")
    dep.__path__ = [os.path.join(SRC, "deprecated")]
    sys.modules["loe_speech_recognition.deprecated"] = dep
    sibling = types.ModuleType("loe_speech_recognition.deprecated.mfcc")
    sibling.compute_mfcc = never
    sys.modules["loe_speech_recognition.deprecated.mfcc"] = sibling
    name = "loe_speech_recognition.deprecated.gaussian_mixture_model"
    spec = importlib.util.spec_from_file_location(name, os.path.join(SRC, "deprecated", "gaussian_mixture_model.py"))
    mod = importlib.util.module_from_spec(spec)
    here = os.getcwd()
    stopped = "ran to the end"
    with tempfile.TemporaryDirectory() as empty:
        os.chdir(empty)
        try:
            spec.loader.exec_module(mod)
        except FileNotFoundError as e:              # the script part looks for ./recordings/...: expected
            stopped = f"{type(e).__name__}: {e}"
        finally:
            os.chdir(here)
    return mod.HMM_GMM, stopped


def main():
    HMM_GMM, stopped = load_reference_class()
    rng = np.random.default_rng(909)
    out = {"module_execution_stopped_at": np.array(stopped)}
    for tag, S, M, T in (("a", 6, 4, 24), ("b", 5, 2, 16)):
        w = rng.dirichlet(np.ones(M) * 2, size=S)
        w[2] = np.concatenate([[0.0], w[2, 1:] / w[2, 1:].sum()])          # a zero weight: log w = -inf
        means = (rng.standard_normal((S, M, 39)) * 2).astype(np.float32)
        cov = np.empty((S, M, 39, 39), np.float32)
        for s in range(S):
            for m in range(M):
                a = rng.standard_normal((39, 39)) * rng.uniform(0.3, 3)
                cov[s, m] = (a @ a.T / 39 + 1e-2 * np.eye(39)).astype(np.float32)
        x = (rng.standard_normal((T, 39)) * 2).astype(np.float32)
        hmm = HMM_GMM(num_states=S, feature_dim=39, num_mixtures=M)
        hmm.mixture_weights = w
        hmm.means = means.astype(np.float64)
        hmm.covariances = cov.astype(np.float64)
        lp = np.empty((T, S))
        with np.errstate(divide="ignore"):
            for t in range(T):
                for s in range(S):
                    lp[t, s] = hmm.log_emission_prob(x[t], s)
        out.update({f"{tag}_weights": w, f"{tag}_means": means, f"{tag}_cov": cov, f"{tag}_x": x, f"{tag}_logprob": lp})
    path = os.path.join(ROOT, "tests", "golden", "gmm_g9_emission.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; module execution stopped at:", stopped)


if __name__ == "__main__":
    main()
