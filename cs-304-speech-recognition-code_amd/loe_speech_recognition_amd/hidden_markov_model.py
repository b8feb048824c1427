"""Gaussian-HMM word models, decoders and trainers with the reference's interface
(src/loe_speech_recognition/hidden_markov_model.py), backed by liblsr_hip.

Every float computation of the hot path (emission scores, Viterbi recursions, label
extraction, alignment statistics) runs on the GPU through the C ABI; this module holds the
model bookkeeping, builds the flat arrays the device consumes and turns reduced statistics
into new parameters.  Each class also offers batched entry points (``predict_batch`` ...)
that replace the reference's ProcessPoolExecutor fan-out with one device call.
"""
from __future__ import annotations

import logging
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import scipy.linalg

from . import _native
from .distributed import all_reduce_stats
from .model_boundary import ModelBoundary
from .signal import Signal, SortedSignals
from .transition_probability import LogTransitionProbabilities, TransitionProbabilities

try:                                     # progress bars are optional
    from tqdm import tqdm
except Exception:                        # pragma: no cover
    class tqdm:                          # type: ignore
        def __init__(self, *a, **k): pass
        def update(self, *a): pass
        def close(self): pass

logger = logging.getLogger(__name__)

_LOG_2PI = float(np.log(2 * np.pi))
MODEL_FILE = "model.npz"


@dataclass
class MultivariateNormal:
    """Frozen full-covariance Gaussian (hidden_markov_model.py:20-48).  Holds what SciPy's frozen
    multivariate_normal holds for ``allow_singular=False``: the float64 mean, the whitening
    matrix U = V diag(lambda^-1/2) of the eigendecomposition and log|Sigma|."""
    dim_of_features: int = field(init=False)
    mean: np.ndarray = field(init=False)
    whiten: np.ndarray = field(init=False)
    log_pdet: float = field(init=False)
    # lower-triangular L with L L^T = U U^T (the form the device multiplies by), when the factorisation that made
    # this object also produced it; otherwise the engine derives it from ``whiten`` at upload time
    tri: Optional[np.ndarray] = field(init=False, default=None, repr=False, compare=False)

    @classmethod
    def from_means_covariances(cls, mean: np.ndarray, covariance: np.ndarray) -> "MultivariateNormal":
        mn = cls()
        mn.mean = np.asarray(mean, dtype=float)
        cov = np.asarray(covariance, dtype=float)
        lam, vec = scipy.linalg.eigh(cov, lower=True, check_finite=True)
        eps = 1e6 * np.finfo("d").eps * np.max(abs(lam))
        if np.min(lam) < -eps:
            raise ValueError("The input matrix must be symmetric positive semidefinite.")
        kept = lam[lam > eps]
        if len(kept) < len(lam):
            raise np.linalg.LinAlgError(
                "When `allow_singular is False`, the input matrix must be symmetric positive definite.")
        mn.whiten = np.multiply(vec, np.sqrt(1.0 / lam))
        mn.log_pdet = float(np.sum(np.log(kept)))
        mn.dim_of_features = mn.mean.shape[0]
        return mn

    @property
    def log_norm(self) -> float:
        return self.dim_of_features * _LOG_2PI + self.log_pdet

    def log_pdf(self, x: np.ndarray) -> np.float32:
        assert x.shape[0] == self.dim_of_features
        eng = _native.get_engine()
        eng.set_model(None, self.mean[None], self.whiten[None], np.array([self.log_norm]))   # single Gaussian
        fb = eng.feature_batch([np.asarray(x, dtype=np.float32)[None, :]])
        return eng.emission(fb)[0, 0].cpu().numpy()[()]


@dataclass
class GaussianMixture:
    """State emission with M weighted full-covariance components: log sum_m w_m N(x; mu_m, Sigma_m),
    accumulated with np.logaddexp semantics in float64 like HMM_GMM.log_emission_prob of the
    reference's retired GMM code (deprecated/gaussian_mixture_model.py:159-164), rounded to float32
    like the live MultivariateNormal.  Can stand wherever a MultivariateNormal stands in a model's
    ``_multivariate_normals`` list (all states of a model set must use the same M)."""
    weights: np.ndarray
    components: List[MultivariateNormal]

    @classmethod
    def from_weights_means_covariances(cls, weights, means, covariances) -> "GaussianMixture":
        comps = [MultivariateNormal.from_means_covariances(m, c) for m, c in zip(means, covariances)]
        w = np.asarray(weights, dtype=np.float64)
        assert w.shape == (len(comps),)
        return cls(w, comps)

    @property
    def dim_of_features(self) -> int:
        return self.components[0].dim_of_features

    def log_pdf(self, x: np.ndarray) -> np.float32:
        assert x.shape[0] == self.dim_of_features
        eng = _native.get_engine()
        eng.set_model(None, *stack_gaussians([self]))
        fb = eng.feature_batch([np.asarray(x, dtype=np.float32)[None, :]])
        return eng.emission(fb)[0, 0].cpu().numpy()[()]


def _host_threads() -> int:
    """Threads for host-side batch work: the job's CPU share (OMP_NUM_THREADS on a shared machine, where
    os.cpu_count() shows every CPU), at most 16."""
    n = os.cpu_count() or 1
    v = os.environ.get("OMP_NUM_THREADS", "")
    if v.isdigit() and int(v) > 0:
        n = min(n, int(v))
    return max(1, min(n, 16))


_factor_executor = None
_second_lapack = False          # False: not looked for yet; None: there is none (or it failed its check)


def _find_second_lapack():
    """(dsyevr address, dgeqrf address, 1) of the ILP64 OpenBLAS build that NumPy's wheel brings along, if it is there
    and factors test matrices to the very bits of SciPy's library; else None.

    Why a second library at all: inside one process the factorisation threads stop scaling at about four, whatever
    the machine (sixteen single-threaded PROCESSES factor the same batch four times faster than sixteen threads),
    because OpenBLAS takes a library-wide lock around the work buffer of nearly every BLAS call.  A second copy of
    the library is a second lock: with the worker threads alternating between the two, eight threads do what four
    did per library.  Both copies are builds of the same OpenBLAS; that they round alike is not taken on trust but
    checked here once per process, and any difference (or a missing symbol) leaves the second copy unused."""
    global _second_lapack
    if _second_lapack is not False:
        return _second_lapack
    _second_lapack = None
    try:
        import ctypes
        import glob

        import scipy.linalg.cython_lapack as lapack

        from . import _lsr_pyhelper as H

        folder = os.path.join(os.path.dirname(os.path.dirname(np.__file__)), "numpy.libs")
        for path in sorted(glob.glob(os.path.join(folder, "libscipy_openblas64_*.so"))):
            lib = ctypes.CDLL(path)                          # NumPy has it loaded already: this only takes a handle
            second = (ctypes.cast(lib.scipy_dsyevr_64_, ctypes.c_void_p).value,
                      ctypes.cast(lib.scipy_dgeqrf_64_, ctypes.c_void_p).value, 1)
            caps = lapack.__pyx_capi__
            rng = np.random.default_rng(7)
            S, d = 12, DIM_CHECK
            a = rng.standard_normal((S, d, d)) * rng.uniform(0.2, 20, size=(S, 1, 1))
            cov = np.ascontiguousarray((a @ a.transpose(0, 2, 1) / d + 1e-3 * np.eye(d)).astype(np.float32), dtype=np.float64)
            out = []
            for extra in ((), second):
                lam, u, l, st = np.zeros((S, d)), np.zeros((S, d, d)), np.zeros((S, d, d)), np.zeros(S, np.int32)
                # two threads: thread 0 uses the first library, thread 1 the second one when it is given
                H.factor_batch(caps["dsyevr"], caps["dgeqrf"], cov, lam, u, l, st, S, d, 0, 1, 2, *extra)
                out.append((lam, u, l, st))
            if all(np.array_equal(x, y) for x, y in zip(*out)) and not out[1][3].any():
                _second_lapack = second
            break
    except Exception:                                        # no such library, other symbol names, ...: one library
        _second_lapack = None
    return _second_lapack


DIM_CHECK = 39


def factor_covariances(means: np.ndarray, covariances: np.ndarray, threads: Optional[int] = None,
                       shard: Optional[Tuple[int, int]] = None, meanwhile=None) -> List[MultivariateNormal]:
    """``[MultivariateNormal.from_means_covariances(m, c) for m, c in zip(means, covariances)]`` as one batch
    (hidden_markov_model.py:283-292 builds one frozen SciPy distribution per state, one after the other).

    The eigendecompositions run on a pool of threads inside the helper extension, through the very LAPACK
    routine scipy.linalg.eigh calls (``dsyevr`` with the workspace sizes of its own query, taken from
    ``scipy.linalg.cython_lapack``), so eigenvalues and whitening matrices carry the bits of the serial path;
    scipy.linalg.eigh itself holds the GIL for the whole LAPACK call and cannot be spread over threads.  The
    triangular factor the device uses (``numpy.linalg.qr`` of the transposed whitening matrix) comes out of
    the same pass.  Errors are raised for the first failing state, with the serial path's exception types.

    ``shard = (rank, world)``: this process factors states rank, rank + world, ... and the results of all
    processes are exchanged with one all-gather (every process must make the same call).
    ``meanwhile``: a callable run on the calling thread while the factorisation is under way."""
    import scipy.linalg.cython_lapack as lapack

    from . import _lsr_pyhelper as H          # built by the Makefile next to this file

    cov = np.ascontiguousarray(covariances, dtype=np.float64)
    means = np.asarray(means, dtype=np.float64)
    S, d = means.shape
    assert cov.shape == (S, d, d)
    finite = np.isfinite(cov).all(axis=(1, 2))
    first, step = (0, 1) if shard is None or shard[1] <= 1 else shard
    lam = np.zeros((S, d))
    whiten = np.zeros((S, d, d))
    tri = np.zeros((S, d, d))
    status = np.zeros(S, dtype=np.int32)
    # a matrix with NaN or infinite entries never reaches LAPACK (scipy.linalg.eigh(check_finite=True))
    safe = cov if finite.all() else np.where(finite[:, None, None], cov, np.eye(d))
    caps = lapack.__pyx_capi__
    n_own = len(range(first, S, step))
    second = _find_second_lapack() if n_own >= 8 and (threads is None or threads >= 4) else None
    if threads is None:                       # one library feeds about four threads (see _find_second_lapack)
        threads = min(_host_threads(), 8 if second else 6)
    args = (caps["dsyevr"], caps["dgeqrf"], np.ascontiguousarray(safe), lam, whiten, tri, status, S, d, first, step,
            threads) + (second or ())
    if meanwhile is None:
        H.factor_batch(*args)
    else:
        # the extension releases the GIL for the whole factorisation: the caller's Python work runs beside it
        global _factor_executor
        if _factor_executor is None:
            from concurrent.futures import ThreadPoolExecutor

            _factor_executor = ThreadPoolExecutor(max_workers=1, thread_name_prefix="lsr-factor")
        pending = _factor_executor.submit(H.factor_batch, *args)
        try:
            meanwhile()
        finally:
            pending.result()
    if step > 1:
        from .distributed import all_gather_strided

        packed = np.concatenate([lam, whiten.reshape(S, -1), tri.reshape(S, -1), status[:, None].astype(np.float64)], axis=1)
        packed = all_gather_strided(packed, first, step)
        lam, whiten = packed[:, :d], packed[:, d:d + d * d].reshape(S, d, d)
        tri, status = packed[:, d + d * d:d + 2 * d * d].reshape(S, d, d), packed[:, -1].astype(np.int32)
    with np.errstate(invalid="ignore", divide="ignore"):
        eps = 1e6 * np.finfo("d").eps * np.max(np.abs(lam), axis=1)
        log_pdet = np.sum(np.log(lam), axis=1)
        bad = ~finite | (status != 0) | ~(lam > eps[:, None]).all(axis=1)
    if bad.any():                     # the first failing state decides, with the error the serial path raises for it
        s = int(np.argmax(bad))
        if not finite[s]:
            raise ValueError("array must not contain infs or NaNs")
        if status[s] != 0:
            raise np.linalg.LinAlgError(f"LAPACK reported an error in the eigendecomposition of state {s} (info {status[s]})")
        if np.min(lam[s]) < -eps[s]:
            raise ValueError("The input matrix must be symmetric positive semidefinite.")
        raise np.linalg.LinAlgError(
            "When `allow_singular is False`, the input matrix must be symmetric positive definite.")
    out = _GaussianBatch()
    pdet = log_pdet.tolist()
    for s in range(S):
        g = MultivariateNormal()
        g.mean, g.whiten, g.log_pdet, g.dim_of_features, g.tri = means[s], whiten[s], pdet[s], d, tri[s]
        out.append(g)
    # the arrays the objects are views of, in the form Engine.set_model takes (see stack_gaussians / stack_rotated)
    out.stacked = (means, whiten, d * _LOG_2PI + log_pdet, None, 1, tri)
    return out


class _GaussianBatch(list):
    """List of MultivariateNormal objects of one factor_covariances call, with the stacked arrays they view."""
    stacked = None


def stack_rotated(gaussians: Sequence) -> Optional[np.ndarray]:
    """Stacked triangular factors of Gaussians that all carry one (see MultivariateNormal.tri), else None."""
    flat = [c for g in gaussians for c in (g.components if isinstance(g, GaussianMixture) else [g])]
    if not flat or any(getattr(g, "tri", None) is None for g in flat):
        return None
    return np.stack([g.tri for g in flat])


def stack_gaussians(gaussians: Sequence) -> Tuple[np.ndarray, np.ndarray, np.ndarray, Optional[np.ndarray], int]:
    """Flat float64 arrays for lsr_model_upload: (mean, whiten, log_norm, log_weights, n_mix).
    Mixtures contribute their components consecutively."""
    n_mix = {len(g.components) if isinstance(g, GaussianMixture) else 1 for g in gaussians}
    if len(n_mix) != 1:
        raise ValueError("all states must have the same number of mixture components")
    n_mix = n_mix.pop()
    flat: List[MultivariateNormal] = []
    logw: List[float] = []
    for g in gaussians:
        if isinstance(g, GaussianMixture):
            flat.extend(g.components)
            with np.errstate(divide="ignore"):
                logw.extend(np.log(g.weights))
        else:
            flat.append(g)
            logw.append(0.0)
    mean = np.stack([g.mean for g in flat])
    whiten = np.stack([g.whiten for g in flat])
    log_norm = np.array([g.log_norm for g in flat], dtype=np.float64)
    return mean, whiten, log_norm, (np.array(logw, dtype=np.float64) if n_mix > 1 else None), n_mix


@dataclass
class HiddenMarkovModel:
    # Mains
    label: str

    # Settings
    isMultiProcessing: bool = field(default=True)     # kept for interface compatibility: batching replaces the pool
    isTqdm: bool = field(default=True)

    # Internals
    _multivariate_normals: List[MultivariateNormal] = field(default_factory=list)
    _log_transition_probs: LogTransitionProbabilities = field(default_factory=LogTransitionProbabilities)

    def __str__(self) -> str:
        return self.label

    @property
    def num_of_states(self) -> int:
        return len(self._multivariate_normals)

    @property
    def dim_of_features(self) -> int:
        return self._multivariate_normals[0].dim_of_features

    # -- device plumbing ---------------------------------------------------------------------------
    def _device_model(self):
        cache = self.__dict__.get("_dev_cache")
        if cache is None or cache["n"] != len(self._multivariate_normals) or cache["src"] is not self._multivariate_normals:
            cache = {"token": _native.new_token(), "n": len(self._multivariate_normals),
                     "src": self._multivariate_normals,
                     "arrays": stack_gaussians(self._multivariate_normals) + (stack_rotated(self._multivariate_normals),)}
            self.__dict__["_dev_cache"] = cache
        return cache["token"], cache["arrays"]

    def __getstate__(self):
        state = dict(self.__dict__)
        state.pop("_dev_cache", None)
        return state

    def _chain(self) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """(state ids, band [3,N], start flags) of this model as one alignment chain."""
        n = self.num_of_states
        start = np.zeros(n, dtype=np.uint8)
        start[0] = 1
        return np.arange(n, dtype=np.int32), self._log_transition_probs.band(), start

    # -- decoding ----------------------------------------------------------------------------------
    def predict(self, signal: np.ndarray) -> Tuple[np.float32, np.ndarray]:
        assert len(self._multivariate_normals) > 0
        assert self.dim_of_features == signal.shape[1]
        return self._viterbi(signal)

    def _viterbi(self, observation_sequence: np.ndarray) -> Tuple[np.float32, np.ndarray]:
        return self.predict_batch([observation_sequence])[0]

    def predict_batch(self, signals, want_paths: bool = True) -> List[Tuple[np.float32, Optional[np.ndarray]]]:
        """(score, int8 path) per utterance from one device call.  ``signals`` is a list of
        (T, 39) arrays or a device-resident FeatureBatch."""
        eng = _native.get_engine()
        fb = signals if isinstance(signals, _native.FeatureBatch) else eng.feature_batch(signals)
        token, arrays = self._device_model()
        eng.set_model(token, *arrays)
        ids, band, start = self._chain()
        d_off, total, d_ids, d_band, d_start = eng.uniform_chains((token, band.tobytes()), fb.n, ids, band, start)
        scores, paths = eng.decode_chain_device(fb, d_off, total, d_ids, d_band, d_start, wide=ids.shape[0] > _native.NARROW_STATES,
                                                want_paths=want_paths)
        scores = scores.cpu().numpy()
        plist = _native.split_ragged(paths.cpu().numpy(), fb.host_off) if want_paths else [None] * fb.n
        return [(scores[i], plist[i]) for i in range(fb.n)]

    # -- persistence (flat arrays instead of the reference's pickled SciPy objects) -----------------
    def _arrays_to_save(self) -> Dict[str, np.ndarray]:
        mean, whiten, _, log_weights, n_mix = stack_gaussians(self._multivariate_normals)
        comps = [c for g in self._multivariate_normals for c in (g.components if isinstance(g, GaussianMixture) else [g])]
        out = {"label": np.array(self.label), "mean": mean, "whiten": whiten,
               "log_pdet": np.array([g.log_pdet for g in comps]),
               "log_trans": self._log_transition_probs.to_array()}
        if n_mix > 1:
            out["mixture_weights"] = np.exp(log_weights).reshape(-1, n_mix)
        return out

    def save(self, parent_folder_path: str = "./cache") -> None:
        folder = os.path.join(parent_folder_path, f"{self.label}")
        os.makedirs(folder, exist_ok=True)
        np.savez(os.path.join(folder, MODEL_FILE), **self._arrays_to_save())

    @classmethod
    def from_folder(cls, model_folder_path: str):
        if not os.path.isdir(model_folder_path):
            raise FileNotFoundError
        model = cls(cls._model_folder_name_parser(model_folder_path))
        with np.load(os.path.join(model_folder_path, MODEL_FILE), allow_pickle=False) as z:
            model._load_arrays(z)
        return model

    def _load_arrays(self, z) -> None:
        comps = []
        for mean, whiten, log_pdet in zip(z["mean"], z["whiten"], z["log_pdet"]):
            g = MultivariateNormal()
            g.mean, g.whiten, g.log_pdet, g.dim_of_features = mean, whiten, float(log_pdet), mean.shape[0]
            comps.append(g)
        if "mixture_weights" in z.files:
            w = z["mixture_weights"]
            m = w.shape[1]
            comps = [GaussianMixture(w[i], comps[i * m:(i + 1) * m]) for i in range(w.shape[0])]
        self._multivariate_normals = comps
        dense = z["log_trans"]
        ltp = LogTransitionProbabilities(dense.shape[0])
        for i in range(dense.shape[0]):
            for j in range(dense.shape[1]):
                if not np.isnan(dense[i, j]):
                    ltp[(i, j)] = dense[i, j]
        self._log_transition_probs = ltp

    @staticmethod
    def _model_folder_name_parser(folder_path: str) -> str:
        return str(folder_path.split("/")[-1])


def _stats_to_host(count, sum_x, sum_xx) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    return count.cpu().numpy(), sum_x.cpu().numpy(), sum_xx.cpu().numpy()


@dataclass
class HiddenMarkovModelTrainable(HiddenMarkovModel):

    class HMMTrainMeanFail(Exception):
        """A state received no frames (hidden_markov_model.py:214-217, 326-329)."""

    class HMMTrainConverge(Exception):
        """New means are np.allclose to the previous ones (hidden_markov_model.py:220-223, 333-335)."""

    _means: np.ndarray = field(init=False)
    _covariances: np.ndarray = field(init=False)
    _transition_probs: TransitionProbabilities = field(init=False)

    @property
    def num_of_states(self) -> int:
        return self._means.shape[0]

    @classmethod
    def from_data(cls, label: str, mfccs: List[np.ndarray], num_of_states: int = 5, max_iterations: int = 100,
                  isMultiProcessingTraining: bool = True, isTqdm: bool = True):
        """Segmental k-means for one word (hidden_markov_model.py:233-281).  The utterances are
        uploaded once; every iteration is one batched alignment plus one statistics reduction.
        ``mfccs`` may also be a FeatureBatch that is already on the device (MFCC.batch_device)."""
        model = cls(label, isMultiProcessing=isMultiProcessingTraining, isTqdm=isTqdm)
        if isinstance(mfccs, _native.FeatureBatch):
            fb = mfccs
            first = fb.feats[:int(fb.lengths[0])].cpu().numpy()
        else:
            fb = None
            first = mfccs[0]
        model._means, model._covariances, model._transition_probs = model._init_parameters(first, num_of_states)
        model._update_inference_weights()
        if fb is None:
            fb = _native.get_engine().feature_batch(mfccs)
        bar = tqdm(desc=f"Train {model.label} model", total=max_iterations, position=1, disable=not model.isTqdm)
        for _ in range(max_iterations):
            try:
                model._train(fb)
                model._update_inference_weights()
            except cls.HMMTrainMeanFail:
                logger.error("Failed to train model")
                raise
            except cls.HMMTrainConverge:
                break
            bar.update()
        bar.close()
        model._update_inference_weights()
        return model

    def _update_inference_weights(self) -> None:
        """hidden_markov_model.py:283-292."""
        self._log_transition_probs = LogTransitionProbabilities.from_transition_probability(self._transition_probs)
        self._multivariate_normals = self.get_multivariate_normals(self._means, self._covariances)

    def _train(self, mfccs) -> None:
        """One iteration (hidden_markov_model.py:294-318): align every utterance with the current
        model, then re-estimate from the frames of each state."""
        eng = _native.get_engine()
        fb = mfccs if isinstance(mfccs, _native.FeatureBatch) else eng.feature_batch(mfccs)
        token, arrays = self._device_model()
        eng.set_model(token, *arrays)
        ids, band, start = self._chain()
        n = ids.shape[0]
        d_off, total, d_ids, d_band, d_start = eng.uniform_chains((token, band.tobytes()), fb.n, ids, band, start)
        _, paths = eng.decode_chain_device(fb, d_off, total, d_ids, d_band, d_start, wide=n > _native.NARROW_STATES)
        # a left-to-right path never decreases, so every frame feeds the state it is aligned to
        # (signal.py:23-47) and consecutive frames of an utterance count as one transition (signal.py:80-91)
        d_off, d_ids = eng.last_chain
        word = eng.torch.zeros(d_ids.shape[0], dtype=eng.torch.int32, device=eng.device)
        slots, trans = eng.remux_slots(paths, fb, d_off, d_ids, word, n, drop_last_run=False)
        count, sum_x, sum_xx = _stats_to_host(*eng.accumulate_stats(fb.feats, slots, fb.total_frames, n))
        self._m_step(count, sum_x, sum_xx, trans.cpu().numpy())

    def _m_step(self, count: np.ndarray, sum_x: np.ndarray, sum_xx: np.ndarray, trans_counts: np.ndarray) -> None:
        """hidden_markov_model.py:320-350 from reduced statistics: mean = sum_x / n,
        cov = (sum_xx - n m m^T) / (n - 1) + 1e-3 I (np.cov's ddof = 1), transitions = counts / row sums."""
        if np.any(count == 0):
            raise self.HMMTrainMeanFail
        mean64 = sum_x / count[:, None]
        new_means = mean64.astype(np.float32)
        if np.allclose(new_means, self._means):
            raise self.HMMTrainConverge
        self._means = new_means
        d = self._covariances.shape[1]
        with np.errstate(invalid="ignore", divide="ignore"):
            n = count[:, None, None]
            cov = (sum_xx - n * (mean64[:, :, None] * mean64[:, None, :])) / (n - 1)
            self._covariances = (cov + np.eye(d) * 0.001).astype(np.float32)
            probs = (trans_counts / np.sum(trans_counts, axis=1, keepdims=True)).astype(np.float32)
        self._transition_probs = TransitionProbabilities.from_transition_probability(probs)

    def _update_middleware_parameters(self, sorted_signals: SortedSignals) -> None:
        """Same re-estimation from host-side aligned signals (hidden_markov_model.py:320-350);
        the frames are uploaded and reduced on the device."""
        n = self.num_of_states
        frames: List[np.ndarray] = []
        slots: List[np.ndarray] = []
        for sig in sorted_signals._signals:
            start = 0
            slot = np.full(len(sig.path), -1, dtype=np.int32)
            for state in range(n):                     # signal.py:23-47 walk
                end = start
                while end < len(sig.path) and sig.path[end] == state:
                    end += 1
                slot[start:end] = state
                start = end
            frames.append(np.asarray(sig.signal, dtype=np.float32))
            slots.append(slot)
        if not frames:
            raise self.HMMTrainMeanFail
        eng = _native.get_engine()
        flat = np.concatenate(frames, axis=0)
        count, sum_x, sum_xx = _stats_to_host(*eng.accumulate_stats(eng.to_device(flat), eng.to_device(np.concatenate(slots)),
                                                                     flat.shape[0], n))
        self._m_step(count, sum_x, sum_xx, sorted_signals.transition_counts)

    def _train_external(self, signals: List[Signal]) -> None:
        sorted_signals = SortedSignals(self.num_of_states)
        for s in signals:
            sorted_signals.append(s)
        self._update_middleware_parameters(sorted_signals)

    @classmethod
    def _init_parameters(cls, sample_signal: np.ndarray, num_of_states: int
                         ) -> Tuple[np.ndarray, np.ndarray, TransitionProbabilities]:
        """hidden_markov_model.py:359-389: means from a uniform split of the FIRST utterance,
        covariances 0.01 I, uniform upper-triangular transitions."""
        seg = int(sample_signal.shape[0] / num_of_states)
        means = [np.average(sample_signal[i * seg: i * seg + seg, :], axis=0) for i in range(num_of_states)]
        return (np.array(means, dtype=np.float32), cls._init_covariance(sample_signal.shape[1], num_of_states),
                TransitionProbabilities.from_num_of_states(num_of_states))

    @staticmethod
    def _init_covariance(dim_of_features: int, num_of_states: int) -> np.ndarray:
        return (np.tile(np.eye(dim_of_features), (num_of_states, 1, 1)) * 0.01).astype(np.float32)

    @staticmethod
    def get_multivariate_normals(means: np.ndarray, covariances: np.ndarray) -> List[MultivariateNormal]:
        return factor_covariances(means, covariances)

    def _arrays_to_save(self) -> Dict[str, np.ndarray]:
        out = super()._arrays_to_save()
        if hasattr(self, "_means"):
            out.update(means=self._means, covariances=self._covariances, trans=self._transition_probs.to_array())
        return out

    def _load_arrays(self, z) -> None:
        super()._load_arrays(z)
        if "means" in z.files:
            self._means, self._covariances = z["means"], z["covariances"]
            self._transition_probs = TransitionProbabilities.from_transition_probability(z["trans"])


def _stack(models: Sequence[HiddenMarkovModel]) -> Tuple[LogTransitionProbabilities, List[MultivariateNormal], ModelBoundary]:
    ltp = LogTransitionProbabilities()
    mvn: List[MultivariateNormal] = []
    mb = ModelBoundary()
    for m in models:
        ltp.append(m._log_transition_probs)
        mvn.extend(m._multivariate_normals)
        mb.append(len(m._multivariate_normals))
    mb.add_model_labels([str(m) for m in models])
    return ltp, mvn, mb


@dataclass
class HiddenMarkovModelInference:
    """Connected-word decoder over a free word loop (hidden_markov_model.py:413-581)."""

    _multivariate_normals: List[MultivariateNormal] = field(default_factory=list)
    _log_transition_probs: LogTransitionProbabilities = field(init=False)
    _model_boundaries: ModelBoundary = field(init=False)
    _log_transition_probability_between_words: float = field(default=np.log(0.005))

    @classmethod
    def from_folder(cls, folder_path: str, models_to_load: List[str]):
        """Word models are stacked in sorted(os.listdir) order (hidden_markov_model.py:421-456)."""
        models = []
        for name in sorted(os.listdir(folder_path)):
            path = os.path.join(folder_path, name)
            if HiddenMarkovModel._model_folder_name_parser(path) in models_to_load:
                models.append(HiddenMarkovModel.from_folder(path))
        return cls.from_models(models)

    @classmethod
    def from_models(cls, models: Sequence[HiddenMarkovModel]):
        inf = cls()
        inf._log_transition_probs, inf._multivariate_normals, inf._model_boundaries = _stack(models)
        return inf

    def __getstate__(self):
        state = dict(self.__dict__)
        state.pop("_dev_cache", None)
        return state

    def _bind(self):
        """Upload model + loop graph if this object's arrays are not the ones on the device."""
        eng = _native.get_engine()
        cache = self.__dict__.get("_dev_cache")
        if cache is None or cache["src"] is not self._multivariate_normals:
            mb = self._model_boundaries
            lower, upper = mb.lower_boundaries, mb.upper_boundaries
            lower_of = np.zeros(mb.num_of_states, dtype=np.int64)
            for lo, up in zip(lower, upper):
                lower_of[lo:up + 1] = lo
            silence = mb._labels.index("S") if "S" in mb._labels else -1
            cache = {"token": _native.new_token(), "src": self._multivariate_normals,
                     "model": stack_gaussians(self._multivariate_normals),
                     "graph": (self._log_transition_probs.band(lower_of), lower, upper, silence)}
            self.__dict__["_dev_cache"] = cache
        eng.set_model(cache["token"], *cache["model"])
        eng.set_loop_graph(cache["token"], *cache["graph"])
        eng.set_between(self._log_transition_probability_between_words)
        return eng

    def _labels_from_device(self, eng, fb, labels, n_labels, label_cap: int) -> List[str]:
        names = self._model_boundaries._labels
        lab = labels.cpu().numpy()
        cnt = n_labels.cpu().numpy()
        out = _native.records_to_strings(lab, cnt, names)
        over = np.nonzero(cnt > label_cap)[0]
        if over.shape[0]:                       # more words than the fixed-width record holds: re-read from the path
            _, paths, _, _ = eng.decode_loop(fb, want_paths=True, label_cap=label_cap)
            plist = _native.split_ragged(paths.cpu().numpy(), fb.host_off)
            for i in over:
                out[i] = "".join(self._model_boundaries.get_labels(plist[i]))
        return out

    def predict(self, signal: np.ndarray) -> str:
        return self.predict_batch([signal])[0]

    def predict_batch(self, signals, label_cap: int = 32) -> List[str]:
        """Decoded label strings for a list of (T, 39) arrays (or a FeatureBatch) in one device call."""
        eng = self._bind()
        fb = signals if isinstance(signals, _native.FeatureBatch) else eng.feature_batch(signals)
        if fb.lengths.min() < 2:
            raise Exception("Failed to find lower boundary for state -1")     # reference: get_labels on path [-1]
        _, _, labels, n_labels = eng.decode_loop(fb, want_paths=False, label_cap=label_cap)
        return self._labels_from_device(eng, fb, labels, n_labels, label_cap)

    def predict_sweep(self, signals, betweens: Sequence, label_cap: int = 32) -> List[List[str]]:
        """Label strings of every utterance for every candidate between-word log-probability — the loop of
        scripts/project5_find_trans_ndigits_with_sil.py:71-88, which re-decodes the same utterances after
        assigning ``_log_transition_probability_between_words`` — with the emission scores computed once.
        ``out[i][b]`` is what ``predict(signals[b])`` returns with the penalty set to ``betweens[i]``."""
        eng = self._bind()
        fb = signals if isinstance(signals, _native.FeatureBatch) else eng.feature_batch(signals)
        if fb.lengths.min() < 2:
            raise Exception("Failed to find lower boundary for state -1")
        _, labels, n_labels = eng.decode_loop_sweep(fb, betweens, label_cap=label_cap)
        names = self._model_boundaries._labels
        lab, cnt = labels.cpu().numpy(), n_labels.cpu().numpy()
        out = []
        for i, between in enumerate(betweens):
            row = _native.records_to_strings(lab[i], cnt[i], names)
            if (cnt[i] > label_cap).any():      # more words than a record holds: decode that penalty alone, from paths
                saved = self._log_transition_probability_between_words
                self._log_transition_probability_between_words = between
                try:
                    row = self.predict_batch(fb, label_cap=label_cap)
                finally:
                    self._log_transition_probability_between_words = saved
            out.append(row)
        return out

    STREAM_FROM_SAMPLES = 4 << 20        # host batches of at least this many samples go through the upload pipeline

    def recognize_audio(self, signals, label_cap: int = 32, streamed: Optional[bool] = None, chunk_samples: int = 0,
                        threads: int = 0) -> List[str]:
        """Audio in, label strings out: MFCC + decode without the features leaving the device
        (the per-utterance work of scripts/project5_test_ndigits_with_sil.py:64-77).
        ``streamed=True`` takes host arrays (int16 as read from a wav file, or float32) through the
        chunked upload pipeline: pinned staging, asynchronous copies overlapped with the decoding of
        earlier chunks; the strings are the same as with one resident batch.  By default (``None``) host
        batches of 4 Mi samples or more are streamed and smaller ones, where one synchronous copy is the
        shorter way, are not."""
        eng = self._bind()
        names = self._model_boundaries._labels
        on_host = not isinstance(signals, _native.AudioBatch)
        if streamed is None:
            streamed = on_host and sum(int(np.shape(s)[0]) for s in signals) >= self.STREAM_FROM_SAMPLES
        if streamed and on_host:
            _, lab, cnt = eng.recognize_host(signals, label_cap=label_cap, chunk_samples=chunk_samples, threads=threads)
            if (cnt > label_cap).any():
                return self.recognize_audio(signals, label_cap=label_cap, streamed=False)
            return _native.records_to_strings(lab, cnt, names)
        ab = signals if not on_host else eng.audio_batch(signals)
        _, _, labels, n_labels = eng.recognize(ab, want_paths=False, label_cap=label_cap)
        cnt = n_labels.cpu().numpy()
        if (cnt > label_cap).any():
            return self.predict_batch(eng.mfcc(ab), label_cap=label_cap)
        return _native.records_to_strings(labels.cpu().numpy(), cnt, names)

    def _viterbi(self, observation_sequence: np.ndarray) -> Tuple[np.float32, np.ndarray]:
        return self.viterbi_batch([observation_sequence])[0]

    def viterbi_batch(self, signals) -> List[Tuple[np.float32, np.ndarray]]:
        eng = self._bind()
        fb = signals if isinstance(signals, _native.FeatureBatch) else eng.feature_batch(signals)
        scores, paths, _, _ = eng.decode_loop(fb, want_paths=True)
        scores = scores.cpu().numpy()
        plist = _native.split_ragged(paths.cpu().numpy(), fb.host_off)
        return [(scores[i], plist[i]) for i in range(fb.n)]


@dataclass
class HiddenMarkovModelMultiWord(HiddenMarkovModel):
    """Forced-alignment chain for a known label string (hidden_markov_model.py:584-664)."""
    _model_boundaries: ModelBoundary = field(init=False)

    @classmethod
    def from_labels(cls, labels: str, trainable_models: Dict[str, HiddenMarkovModelTrainable]):
        hmm = cls(labels)
        hmm.isTqdm = False
        hmm._log_transition_probs, hmm._multivariate_normals, hmm._model_boundaries = _stack(
            [_Labelled(lab, trainable_models[lab]) for lab in labels])
        return hmm

    def get_remuexed_signals(self, mfccs_sequences: List[np.ndarray]) -> Dict[str, List[Signal]]:
        out: Dict[str, List[Signal]] = {lab: [] for lab in self._model_boundaries._labels}
        for x, (_, path) in zip(mfccs_sequences, self.predict_batch(mfccs_sequences)):
            for lab, sigs in self._remux_path_and_signal(x, path, self._model_boundaries).items():
                out[lab].extend(sigs)
        return out

    @staticmethod
    def _remux_path_and_signal(signal: np.ndarray, path: np.ndarray, model_boundaries: ModelBoundary
                               ) -> Dict[str, List[Signal]]:
        """hidden_markov_model.py:602-636: cut wherever the word label changes; a piece is only
        emitted by the change that ends it, so the final piece is dropped."""
        out: Dict[str, List[Signal]] = {lab: [] for lab in model_boundaries._labels}
        begin = 0
        cur_state = int(path[0])
        cur_label = model_boundaries.get_label(cur_state)
        for i, st in enumerate(path):
            lab = model_boundaries.get_label(int(st))
            if lab != cur_label:
                lo = model_boundaries.find_lower_boundary(cur_state)
                hi = model_boundaries.find_upper_boundary(cur_state)
                out[cur_label].append(Signal(num_of_state=hi - lo + 1, signal=signal[begin:i], path=path[begin:i] - lo))
                begin, cur_state = i, int(path[i])
                cur_label = model_boundaries.get_label(cur_state)
        return out


class _Labelled:
    """Adapter: a word model seen under the label it has in a chain."""

    def __init__(self, label: str, model: HiddenMarkovModel):
        self._label = label
        self._log_transition_probs = model._log_transition_probs
        self._multivariate_normals = model._multivariate_normals

    def __str__(self) -> str:
        return self._label


@dataclass
class HiddenMarkovModelTrainContinuous:
    """Continuous-speech re-training of all word models (hidden_markov_model.py:667-797)."""
    isTqdm: bool = field(default=True)
    isMultiProcessing: bool = field(default=True)     # interface compatibility; the batch replaces the pool

    _trainable_models: Dict[str, HiddenMarkovModelTrainable] = field(default_factory=dict)
    _models_loaded: List[str] = field(default_factory=list)
    _num_of_finished_models: int = field(default=0)
    # with several ranks: deal the per-state factorisation of the M-step out over the ranks (else every rank
    # factors every state)
    shard_factorization: bool = field(default=True)

    @classmethod
    def from_folder(cls, folder_path: str, models_to_load: List[str]):
        obj = cls()
        for name in sorted(os.listdir(folder_path)):
            path = os.path.join(folder_path, name)
            label = HiddenMarkovModel._model_folder_name_parser(path)
            if label not in models_to_load:
                continue
            obj.add_model(HiddenMarkovModelTrainable.from_folder(path))
        obj._models_loaded = models_to_load
        return obj

    def __getstate__(self):
        # device buffers and what was derived from them stay behind: a copy rebuilds them on first use
        state = dict(self.__dict__)
        for key in ("_data_cache", "_chain_cache", "_stack_cache", "_factored", "_factored_states"):
            state.pop(key, None)
        return state

    def add_model(self, m: HiddenMarkovModelTrainable) -> None:
        """Register a word model the way from_folder does (hidden_markov_model.py:692-707): the
        inference weights are kept, the trainable parameters restart from zeros / 0.01 I /
        uniform so that the first iteration can never report convergence."""
        n = len(m._multivariate_normals)
        d = m._multivariate_normals[0].dim_of_features
        m._means = np.zeros((n, d), dtype=np.float32)
        m._covariances = m._init_covariance(d, n)
        m._transition_probs = TransitionProbabilities.from_num_of_states(n)
        self._trainable_models[m.label] = m
        if m.label not in self._models_loaded:
            self._models_loaded.append(m.label)

    # -- one iteration on the device --------------------------------------------------------------------
    def _layout(self) -> Tuple[Dict[str, int], int]:
        base, n = {}, 0
        for lab in self._models_loaded:
            base[lab] = n
            n += self._trainable_models[lab].num_of_states
        return base, n

    def _prepare(self, labeled_mfccs: Optional[Dict[str, List[np.ndarray]]]):
        """Upload the utterances once; remember which label string each one carries.  The upload is reused
        only for the very same dict holding the very same arrays (the cache keeps both alive, so an
        address cannot be recycled by new data); ``None`` selects a batch installed with use_device_batch."""
        cache = self.__dict__.get("_data_cache")
        if labeled_mfccs is None:
            if cache is None or cache["src"] is not None:
                raise ValueError("no device batch installed: call use_device_batch(features, label_strings) first")
            return cache["fb"], cache["owner"]
        finger = tuple((digits, len(seqs), tuple(map(id, seqs))) for digits, seqs in labeled_mfccs.items())
        if cache is None or cache["src"] is not labeled_mfccs or cache["finger"] != finger:
            eng = _native.get_engine()
            utts, owner = [], []
            for digits, seqs in labeled_mfccs.items():
                utts.extend(seqs)
                owner.extend([digits] * len(seqs))
            cache = {"src": labeled_mfccs, "finger": finger, "keep": utts, "fb": eng.feature_batch(utts), "owner": owner}
            self.__dict__["_data_cache"] = cache
        return cache["fb"], cache["owner"]

    def use_device_batch(self, fb: "_native.FeatureBatch", label_strings: Sequence[str]) -> None:
        """Train on features that are already on the device (e.g. from MFCC.batch_device): utterance b of
        ``fb`` carries the digit string ``label_strings[b]``.  Afterwards ``train(None)`` / ``_iteration(None)``
        use this batch."""
        if fb.n != len(label_strings):
            raise ValueError("one label string per utterance")
        self.__dict__["_data_cache"] = {"src": None, "finger": None, "keep": None, "fb": fb, "owner": list(label_strings)}

    def _chains(self, owner: List[str]):
        """Flat arrays describing the alignment chain S d S d ... S of every utterance
        (hidden_markov_model.py:638-664, 794-797), built without per-utterance Python work:
        chain offsets, model state of every chain position, its 3-band transitions (keys that
        cross a word boundary are missing in the reference's stacked store and read 0.0),
        start flags and the ordinal of the word each position belongs to."""
        base, n_total = self._layout()
        labels = self._models_loaded
        model_band = np.concatenate([self._trainable_models[lab]._log_transition_probs.band() for lab in labels], axis=1)
        cache = self.__dict__.get("_chain_cache")
        layout = tuple((lab, self._trainable_models[lab].num_of_states) for lab in labels)
        # the structure only depends on the label strings and on the order and sizes of the models
        if cache is not None and cache["owner"] is owner and cache["layout"] == layout:
            off, ids, start, ordinal, cross1, cross2 = cache["arrays"]
            band = model_band[:, ids]
            band[1, cross1] = 0.0
            band[2, cross2] = 0.0
            return off, ids, band.astype(np.float32), start, ordinal, n_total
        sizes = np.array([self._trainable_models[lab].num_of_states for lab in labels], dtype=np.int64)
        bases = np.array([base[lab] for lab in labels], dtype=np.int64)
        lut = np.full(256, -1, dtype=np.int64)
        for i, lab in enumerate(labels):
            lut[ord(lab)] = i
        text = "".join(self.insert_silence(d) for d in owner)
        word = lut[np.frombuffer(text.encode("ascii"), dtype=np.uint8)]
        if (word < 0).any():
            raise KeyError("label string contains a word without a model")
        words_per_utt = np.array([2 * len(d) + 1 for d in owner], dtype=np.int64)
        utt_of_word = np.repeat(np.arange(len(owner)), words_per_utt)
        first_word = np.concatenate([[0], np.cumsum(words_per_utt)[:-1]])
        ordinal_of_word = np.arange(word.shape[0]) - first_word[utt_of_word]
        wsize = sizes[word]
        wstart = np.concatenate([[0], np.cumsum(wsize)[:-1]])
        total = int(wsize.sum())
        pos_in_word = np.arange(total) - np.repeat(wstart, wsize)
        ids = np.repeat(bases[word], wsize) + pos_in_word
        ordinal = np.repeat(ordinal_of_word, wsize)
        band = model_band[:, ids].copy()
        later = ordinal > 0                       # not the first word of its utterance
        cross1 = np.nonzero(later & (pos_in_word == 0))[0]
        cross2 = np.nonzero(later & (pos_in_word <= 1))[0]
        band[1, cross1] = 0.0
        band[2, cross2] = 0.0
        chain_len = np.bincount(utt_of_word, weights=wsize, minlength=len(owner)).astype(np.int64)
        off = np.concatenate([[0], np.cumsum(chain_len)]).astype(np.int64)
        start = np.zeros(total, dtype=np.uint8)
        start[off[:-1]] = 1
        ids = ids.astype(np.int32)
        ordinal = ordinal.astype(np.int64)
        self.__dict__["_chain_cache"] = {"owner": owner, "layout": layout, "arrays": (off, ids, start, ordinal, cross1, cross2)}
        return off, ids, band.astype(np.float32), start, ordinal, n_total

    def _align(self, fb, owner: List[str]):
        """Forced alignment of every utterance against S d S d ... S (hidden_markov_model.py:772-782)
        in one device call.  Returns the device paths (chain positions) and the chain description.
        The structure of the chains (offsets, model states, start flags, which band entries cross a word
        boundary) only depends on the label strings and stays on the device between iterations; what changes
        is the 3 x n_states band of the models, which is uploaded and spread over the chain positions there."""
        eng = _native.get_engine()
        t = eng.torch
        models = [self._trainable_models[lab] for lab in self._models_loaded]
        eng.set_model(_native.new_token(), *self._stacked_gaussians(models))
        cache = self.__dict__.get("_chain_cache")
        layout = tuple((lab, self._trainable_models[lab].num_of_states) for lab in self._models_loaded)
        if cache is None or cache["owner"] is not owner or cache["layout"] != layout or "d_ids" not in cache:
            off, ids, _, start, ordinal, n_total = self._chains(owner)
            cache = self.__dict__["_chain_cache"]
            _, _, _, _, cross1, cross2 = cache["arrays"]
            if int(np.diff(off).max()) > _native.MAX_STATES:
                raise ValueError(f"every chain needs 1..{_native.MAX_STATES} states")
            cache.update(d_off=eng.to_device(off), d_ids=eng.to_device(ids), d_ids_long=eng.to_device(ids.astype(np.int64)),
                         d_start=eng.to_device(start), d_cross1=eng.to_device(cross1.astype(np.int64)),
                         d_cross2=eng.to_device(cross2.astype(np.int64)), d_ordinal=eng.to_device(ordinal.astype(np.int32)),
                         wide=int(np.diff(off).max()) > _native.NARROW_STATES, n_total=n_total)
        off, ids, _, ordinal = cache["arrays"][0], cache["arrays"][1], None, cache["arrays"][3]
        model_band = np.concatenate([m._log_transition_probs.band() for m in models], axis=1).astype(np.float32)
        if np.isnan(model_band).any():
            raise ValueError("NaN transition log-probability")
        d_band = eng.to_device(model_band)[:, cache["d_ids_long"]].contiguous()
        d_band[1, cache["d_cross1"]] = 0.0
        d_band[2, cache["d_cross2"]] = 0.0
        _, paths = eng.decode_chain_device(fb, cache["d_off"], int(off[-1]), cache["d_ids"], d_band, cache["d_start"],
                                           wide=cache["wide"])
        return paths, off, ids, ordinal, cache["n_total"]

    def _iteration(self, labeled_mfccs: Dict[str, List[np.ndarray]]) -> None:
        """Alignment + re-mux + statistics + per-word M-step (hidden_markov_model.py:733-770) without
        materialising per-word Signal objects: a frame contributes to the model state it is aligned
        to unless it belongs to the last word run of its utterance (the reference never flushes that
        run, hidden_markov_model.py:614-634).  Paths, slots and statistics stay on the device; one
        packed float64 vector comes back (after the all-reduce when several ranks train together)."""
        eng = _native.get_engine()
        stats, n_total = self._local_statistics(labeled_mfccs)
        stats = all_reduce_stats(stats)
        stats = eng.to_host_pinned(stats)
        d = _native.DIM
        count = stats[:n_total]
        sum_x = stats[n_total:n_total * (1 + d)].reshape(n_total, d)
        sum_xx = stats[n_total * (1 + d):n_total * (1 + d + d * d)].reshape(n_total, d, d)
        trans = stats[n_total * (1 + d + d * d):].reshape(n_total, n_total)
        base, _ = self._layout()
        self._m_steps(base, count, sum_x, sum_xx, trans)

    def _local_statistics(self, labeled_mfccs: Optional[Dict[str, List[np.ndarray]]]):
        """This process's share of an iteration before the exchange: forced alignment of its utterances, re-mux,
        and the packed float64 device vector (count [S], sum_x [S, 39], sum_xx [S, 39, 39], transition counts
        [S, S]) that the ranks add up.  Returns (vector, S)."""
        eng = _native.get_engine()
        t = eng.torch
        fb, owner = self._prepare(labeled_mfccs)
        paths, off, ids, ordinal, n_total = self._align(fb, owner)
        d_off, d_ids = eng.last_chain
        cache = self.__dict__["_chain_cache"]
        if "d_ordinal" not in cache:
            cache["d_ordinal"] = eng.to_device(ordinal.astype(np.int32))
        slots, trans = eng.remux_slots(paths, fb, d_off, d_ids, cache["d_ordinal"], n_total, drop_last_run=True)
        count, sum_x, sum_xx = eng.accumulate_stats(fb.feats, slots, fb.total_frames, n_total)
        return t.cat([count, sum_x.reshape(-1), sum_xx.reshape(-1), trans.reshape(-1).to(t.float64)]), n_total

    def _stacked_gaussians(self, models: Sequence[HiddenMarkovModelTrainable]):
        """Arguments of Engine.set_model for the Gaussians of all word models in layout order.  After an
        iteration of this trainer the stacked arrays of the batch factorisation are still at hand (and
        include the triangular factors); otherwise they are gathered from the models' objects."""
        cache = self.__dict__.get("_stack_cache")
        lists = [m._multivariate_normals for m in models]
        if cache is not None and len(cache["lists"]) == len(lists) and all(a is b for a, b in zip(cache["lists"], lists)):
            return cache["arrays"]
        flat = [g for lst in lists for g in lst]
        return stack_gaussians(flat) + (stack_rotated(flat),)

    def _refresh_inference_weights(self, models: Sequence[HiddenMarkovModelTrainable]) -> None:
        """``_update_inference_weights`` of several word models at once (hidden_markov_model.py:283-292, 766-767):
        the log transition stores per model, the Gaussians of all their states as ONE batch factorisation on
        the host threads - and, when several ranks train together, with the states dealt out over the ranks
        (rank r factors states r, r + world, ...; one all-gather of the factors), so that the host work of an
        iteration divides by the number of ranks like the device work does."""
        if not models:
            return
        # a model that reported convergence kept its parameter arrays: its Gaussians are the ones already there
        # (the reference recomputes them from the unchanged parameters, with the same result)
        done = self.__dict__.setdefault("_factored", {})
        stale = [m for m in models
                 if done.get(m.label, (None,))[0] is not m._means or done[m.label][1] is not m._covariances
                 or done[m.label][2] is not m._multivariate_normals]
        from .distributed import is_active, rank_world

        def log_transitions():
            for m in models:
                m._log_transition_probs = LogTransitionProbabilities.from_transition_probability(m._transition_probs)

        if stale:
            means = np.concatenate([m._means for m in stale], axis=0)
            covs = np.concatenate([m._covariances for m in stale], axis=0)
            shard = rank_world() if self.shard_factorization and is_active() else None
            gaussians = factor_covariances(means, covs, shard=shard, meanwhile=log_transitions)
            lo = 0
            for m in stale:
                m._multivariate_normals = gaussians[lo:lo + m._means.shape[0]]
                lo += m._means.shape[0]
                done[m.label] = (m._means, m._covariances, m._multivariate_normals)
        else:
            log_transitions()
        self.__dict__["_factored_states"] = sum(m._means.shape[0] for m in stale)
        if [m.label for m in models] == list(self._models_loaded):
            if len(stale) == len(models):
                arrays = gaussians.stacked                    # every model was factored: the batch arrays as they are
            else:
                flat = [g for m in models for g in m._multivariate_normals]
                arrays = stack_gaussians(flat) + (stack_rotated(flat),)
            self.__dict__["_stack_cache"] = {"lists": [m._multivariate_normals for m in models], "arrays": arrays}

    def _m_steps(self, base, count, sum_x, sum_xx, trans) -> None:
        """hidden_markov_model.py:754-770, including the never-reset convergence counter: a model whose means
        did not move counts as finished every time, and the iteration that brings the counter to the number of
        models ends the training at that model (the models after it keep their parameters).  The reference
        refreshes a model's inference weights in a ``finally`` right after its update; here the models that
        were reached are refreshed together afterwards (_refresh_inference_weights)."""
        reached: List[HiddenMarkovModelTrainable] = []
        stop: Optional[BaseException] = None
        # the arithmetic of HiddenMarkovModelTrainable._m_step for the states of all models at once (element by
        # element the same operations, so the same bits); what stays per model is the control flow
        models = [self._trainable_models[lab] for lab in self._models_loaded]
        d = sum_x.shape[1]
        with np.errstate(invalid="ignore", divide="ignore"):
            mean64 = sum_x / count[:, None]
            new_means = mean64.astype(np.float32)
            n = count[:, None, None]
            cov = ((sum_xx - n * (mean64[:, :, None] * mean64[:, None, :])) / (n - 1) + np.eye(d) * 0.001).astype(np.float32)
        old_means = np.concatenate([m._means for m in models], axis=0)
        close = np.isclose(new_means, old_means).all(axis=1) if old_means.shape == new_means.shape else None
        for lab, m in zip(self._models_loaded, models):
            lo, hi = base[lab], base[lab] + m.num_of_states
            reached.append(m)
            if close is None:                         # layout and statistics disagree: the per-model form reports it
                m._m_step(count[lo:hi], sum_x[lo:hi], sum_xx[lo:hi], trans[lo:hi, lo:hi])
                continue
            if np.any(count[lo:hi] == 0):
                stop = HiddenMarkovModelTrainable.HMMTrainMeanFail()
                break
            if close[lo:hi].all():                    # np.allclose(new, old): the model keeps all its parameters
                self._num_of_finished_models += 1
                if self._num_of_finished_models == len(self._trainable_models):
                    stop = HiddenMarkovModelTrainable.HMMTrainConverge()
                    break
                continue
            m._means = new_means[lo:hi]
            m._covariances = cov[lo:hi]
            block = trans[lo:hi, lo:hi]
            with np.errstate(invalid="ignore", divide="ignore"):
                probs = (block / np.sum(block, axis=1, keepdims=True)).astype(np.float32)
            m._transition_probs = TransitionProbabilities.from_transition_probability(probs)
        self._refresh_inference_weights(reached)
        if stop is not None:
            raise stop

    def train(self, labeled_mfccs: Dict[str, List[np.ndarray]], max_iterations: int = 100) -> None:
        bar = tqdm(total=max_iterations, desc="Training Iteration", disable=not self.isTqdm, position=0)
        for _ in range(max_iterations):
            try:
                self._iteration(labeled_mfccs)
            except HiddenMarkovModelTrainable.HMMTrainMeanFail:
                logger.error("Failed to train model")
                raise
            except HiddenMarkovModelTrainable.HMMTrainConverge:
                break
            bar.update()
        bar.close()

    # -- reference-shaped two-step interface -------------------------------------------------------------
    def _train(self, labeled_mfccs: Dict[str, List[np.ndarray]]) -> Dict[str, List[Signal]]:
        """Re-muxed per-word Signal lists as the reference returns them (hidden_markov_model.py:733-752)."""
        out: Dict[str, List[Signal]] = {lab: [] for lab in self._models_loaded}
        for item in labeled_mfccs.items():
            for lab, sigs in self._train_process(item).items():
                out[lab].extend(sigs)
        return out

    def _train_process(self, labels_and_mfccs: Tuple[str, List[np.ndarray]]) -> Dict[str, List[Signal]]:
        labels, mfccs = labels_and_mfccs
        hmm = HiddenMarkovModelMultiWord.from_labels(self.insert_silence(labels), self._trainable_models)
        return hmm.get_remuexed_signals(mfccs)

    def _update_trainable_model_parameters(self, remuxed_signals: Dict[str, List[Signal]]) -> None:
        for label, signals in remuxed_signals.items():
            try:
                self._trainable_models[label]._train_external(signals)
            except HiddenMarkovModelTrainable.HMMTrainConverge:
                self._num_of_finished_models += 1
                if self._num_of_finished_models == len(self._trainable_models):
                    raise
            finally:
                self._trainable_models[label]._update_inference_weights()

    def save(self, folder_path: str) -> None:
        os.makedirs(folder_path, exist_ok=True)
        for model in self._trainable_models.values():
            model.save(folder_path)

    @staticmethod
    def insert_silence(labels: str) -> str:
        return "".join(f"S{c}" for c in labels) + "S"
