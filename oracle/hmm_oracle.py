"""CPU oracle for the Gaussian-HMM half of the hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in NumPy/SciPy, what the reference package
``loe_speech_recognition`` computes for Gaussian emission scores, Viterbi
decoding, path -> label conversion and the segmental k-means M-step.  Only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it; the shipped package never does.

Parity status: PINNED.  ``tests/golden/hmm_*.npz`` hold outputs of the real
reference code (imported in the build container by ``tools/make_golden_hmm.py``)
and ``tests/test_oracle_hmm.py`` checks every function below against them.

Two flavours are provided for the numerics:
  * ``*_ref``  functions keep the reference's loop structure (one SciPy frozen
    ``logpdf`` call per (frame, state), scalar NumPy arithmetic with the same
    dtypes).  They are the pinned oracle and the CPU baseline that is timed.
  * ``*_fast`` functions are vectorised restatements with identical results
    (checked against the ``*_ref`` ones) used for larger parity cases.

Reference citations are relative to /root/reference/src/loe_speech_recognition/.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import scipy.linalg
import scipy.stats

NEG_INF = -float("inf")
LOG_2PI = float(np.log(2.0 * np.pi))  # scipy/stats/_multivariate.py: _LOG_2PI


# ----------------------------------------------------------------------------
# Gaussian emission  (hidden_markov_model.py:20-48 + scipy.stats frozen MVN)
# ----------------------------------------------------------------------------
def whitening_from_cov(cov: np.ndarray) -> Tuple[np.ndarray, float, int]:
    """U, log|Sigma|, rank exactly as scipy's ``_PSD`` builds them.

    scipy/stats/_multivariate.py ``_PSD.__init__``: eigh(lower=True) in float64,
    eps = 1e6 * eps_f64 * max|lambda|, U = V * sqrt(1/lambda), logdet = sum log lambda.
    ``allow_singular=False`` (hidden_markov_model.py:36-40) -> LinAlgError when
    any eigenvalue is <= eps.
    """
    cov64 = np.asarray(cov, dtype=np.float64)
    lam, vec = scipy.linalg.eigh(cov64, lower=True, check_finite=True)
    eps = 1e6 * np.finfo(np.float64).eps * np.max(np.abs(lam))
    if np.min(lam) < -eps:
        raise ValueError("The input matrix must be symmetric positive semidefinite.")
    kept = lam[lam > eps]
    if len(kept) < len(lam):
        raise np.linalg.LinAlgError(
            "When `allow_singular is False`, the input matrix must be symmetric positive definite.")
    inv = np.array([0 if abs(v) <= eps else 1 / v for v in lam], dtype=float)
    U = np.multiply(vec, np.sqrt(inv))
    return U, float(np.sum(np.log(kept))), int(len(kept))


class GaussianRef:
    """One state's emission density, evaluated the way the reference does:
    a SciPy frozen multivariate normal, float64 inside, float32 out
    (hidden_markov_model.py:27-48)."""

    def __init__(self, mean: np.ndarray, cov: np.ndarray):
        self.mean32 = np.asarray(mean, dtype=np.float32)
        self.cov32 = np.asarray(cov, dtype=np.float32)
        self.frozen = scipy.stats.multivariate_normal(
            mean=self.mean32, cov=self.cov32, allow_singular=False)
        self.dim = int(self.mean32.shape[0])

    def log_pdf(self, x: np.ndarray) -> np.float32:
        assert x.shape[0] == self.dim
        return self.frozen.logpdf(x).astype(np.float32)

    # flat float64 export: what the device model upload consumes
    def export(self) -> Tuple[np.ndarray, np.ndarray, float]:
        U, logdet, rank = whitening_from_cov(self.cov32)
        assert rank == self.dim
        return self.mean32.astype(np.float64), U, logdet


def emission_ref(x: np.ndarray, gaussians: Sequence[GaussianRef]) -> np.ndarray:
    """float32 [T, S] log-likelihoods, one frozen logpdf call per element."""
    T = x.shape[0]
    out = np.empty((T, len(gaussians)), dtype=np.float32)
    for t in range(T):
        for s, g in enumerate(gaussians):
            out[t, s] = g.log_pdf(x[t])
    return out


def emission_fast(x: np.ndarray, means64: np.ndarray, U: np.ndarray, logdet: np.ndarray) -> np.ndarray:
    """Vectorised float64 restatement of scipy's ``_logpdf``:
    -0.5 * (D*log(2pi) + logdet + sum(((x - mu) @ U)**2)), rounded once to float32.
    x f32[T,D]; means64 f64[S,D]; U f64[S,D,D]; logdet f64[S].
    BLAS summation order differs from the per-call path, so individual float32
    values may differ from ``emission_ref`` by one ulp on rare elements."""
    x64 = np.asarray(x, dtype=np.float64)
    D = x64.shape[1]
    dev = x64[None, :, :] - means64[:, None, :]            # [S,T,D]
    z = np.matmul(dev, U)                                   # [S,T,D]
    maha = np.sum(np.square(z), axis=-1)                    # [S,T]
    ll = -0.5 * ((D * LOG_2PI + logdet)[:, None] + maha)
    return np.ascontiguousarray(ll.T).astype(np.float32)


def emission_gmm_ref(x: np.ndarray, weights: np.ndarray, gaussians: Sequence[Sequence[GaussianRef]]) -> np.ndarray:
    """Mixture emissions the way the reference's retired GMM code forms them
    (deprecated/gaussian_mixture_model.py:159-164): log_prob = -inf, then
    log_prob = np.logaddexp(log_prob, np.log(w_m) + logpdf_m(x)) component by component in
    float64; rounded to float32 like the live code's MultivariateNormal.log_pdf.
    Pinned by tests/golden/gmm_g9_emission.npz: values returned by the reference's own
    HMM_GMM.log_emission_prob (tools/make_golden_gmm.py executes that module, which is dead code in
    the reference and fails in its script part, up to the point where the class exists)."""
    T = x.shape[0]
    out = np.empty((T, len(gaussians)), dtype=np.float32)
    with np.errstate(divide="ignore"):
        for t in range(T):
            for s, comps in enumerate(gaussians):
                lp = -float("inf")
                for m, g in enumerate(comps):
                    lp = np.logaddexp(lp, np.log(weights[s][m]) + g.frozen.logpdf(x[t]))
                out[t, s] = lp
    return out


def emission_gmm_fast(x: np.ndarray, weights: np.ndarray, gaussians: Sequence[Sequence[GaussianRef]]) -> np.ndarray:
    """Vectorised restatement of ``emission_gmm_ref`` (same sequential logaddexp order)."""
    S, M = len(gaussians), len(gaussians[0])
    mu, U, ld = export_gaussians([g for comps in gaussians for g in comps])
    x64 = np.asarray(x, dtype=np.float64)
    D = x64.shape[1]
    dev = x64[None] - mu[:, None, :]
    maha = np.sum(np.square(np.matmul(dev, U)), axis=-1)
    lp = (-0.5 * ((D * LOG_2PI + ld)[:, None] + maha)).reshape(S, M, -1)
    with np.errstate(divide="ignore"):
        lw = np.log(np.asarray(weights, dtype=np.float64))
    acc = np.full((S, x64.shape[0]), -np.inf)
    for m in range(M):
        acc = np.logaddexp(acc, lw[:, m, None] + lp[:, m])
    return np.ascontiguousarray(acc.T).astype(np.float32)


# ----------------------------------------------------------------------------
# Sparse log-transition store  (transition_probability.py:11-82)
# ----------------------------------------------------------------------------
class LogTransRef:
    """(i, j) -> log-probability; a key that was never stored reads as 0.0
    (transition_probability.py:17-23).  ``append`` stacks models block-diagonally
    (transition_probability.py:70-75)."""

    def __init__(self, n: int = 0):
        self.n = n
        self.core: Dict[Tuple[int, int], float] = {}

    def get(self, i: int, j: int):
        return self.core.get((i, j), 0.0)

    @classmethod
    def from_probs(cls, probs: np.ndarray) -> "LogTransRef":
        """Dense N x N float32 probabilities -> log values; zeros become -inf and
        are stored (transition_probability.py:54-64, 77-82)."""
        p = np.asarray(probs, dtype=np.float32)
        obj = cls(p.shape[0])
        with np.errstate(divide="ignore"):
            for i in range(p.shape[0]):
                for j in range(p.shape[1]):
                    if p[i, j] != NEG_INF:
                        obj.core[(i, j)] = np.log(p[i, j])   # np.float32
        return obj

    def append(self, other: "LogTransRef") -> None:
        base = self.n
        self.n += other.n
        for (i, j), v in other.core.items():
            self.core[(i + base, j + base)] = v


def uniform_upper_triangular(n: int) -> np.ndarray:
    """Initial transition matrix: row i spreads 1/(n-i) over columns i..n-1
    (transition_probability.py:42-52)."""
    rows = []
    for i in range(n):
        rows.append([0.0] * i + [1 / (n - i)] * (n - i))
    return np.array(rows, dtype=np.float32)


# ----------------------------------------------------------------------------
# Word boundaries and path -> labels  (model_boundary.py)
# ----------------------------------------------------------------------------
class BoundariesRef:
    def __init__(self, sizes: Sequence[int], labels: Sequence[str]):
        assert len(sizes) == len(labels)                      # model_boundary.py:102-105
        ends = np.cumsum(np.asarray(sizes, dtype=np.int64))
        self.lower = [0] + [int(e) for e in ends[:-1]]        # model_boundary.py:25-39
        self.upper = [int(e) - 1 for e in ends]               # model_boundary.py:41-55
        self.labels = list(labels)
        self.num_words = len(sizes)
        self.num_states = int(ends[-1])

    def find_lower(self, state: int) -> int:                  # model_boundary.py:82-90
        for lb in reversed(self.lower):
            if state >= lb:
                return lb
        raise Exception(f"no lower boundary for state {state}")

    def find_upper(self, state: int) -> int:                  # model_boundary.py:92-100
        for ub in self.upper:
            if state <= ub:
                return ub
        raise Exception(f"no upper boundary for state {state}")

    def label_of(self, state: int) -> str:                    # model_boundary.py:149-163
        return self.labels[self.lower.index(self.find_lower(state))]

    def get_labels(self, path: np.ndarray, skip_silence: bool = True) -> List[str]:
        """model_boundary.py:107-147: run-length compress, emit the first run's
        word, then a word whenever the state leaves the current word's range or
        jumps from the word's last state straight back to its first state."""
        seq = [int(v) for v in np.asarray(path).tolist()]
        runs = [seq[0]]
        for v in seq[1:]:
            if v != runs[-1]:
                runs.append(v)
        out: List[str] = []

        def emit(state: int) -> None:
            lab = self.label_of(state)
            if not (lab == "S" and skip_silence):
                out.append(lab)

        lo, hi = self.find_lower(runs[0]), self.find_upper(runs[0])
        emit(runs[0])
        for i in range(1, len(runs)):
            cur = runs[i]
            if cur < lo or cur > hi:
                lo, hi = self.find_lower(cur), self.find_upper(cur)
                emit(cur)
            elif runs[i - 1] == hi and cur == lo:
                emit(cur)
        return out


# ----------------------------------------------------------------------------
# Viterbi, reference loop structure  (hidden_markov_model.py:80-91,160-208,463-581)
# ----------------------------------------------------------------------------
def _backtrace(tracer: np.ndarray, terminal: int) -> np.ndarray:
    """hidden_markov_model.py:200-208 / 573-581.  The terminal state itself is
    never written: path[T-1] repeats the predecessor chosen at the last frame."""
    T = tracer.shape[0]
    prev_state = tracer[-1, terminal]
    path = np.zeros((T,), dtype=np.int8)
    path[-1] = prev_state
    for t in range(T - 2, -1, -1):
        path[t] = prev_state
        prev_state = tracer[t, prev_state]
    return path


def viterbi_chain_ref(x: np.ndarray, ltp: LogTransRef, gaussians: Sequence[GaussianRef]
                      ) -> Tuple[np.float32, np.ndarray]:
    """Single left-to-right model or stacked alignment chain
    (hidden_markov_model.py:80-91 init, 160-208 recursion)."""
    N = len(gaussians)
    T = x.shape[0]
    left = np.full((N,), NEG_INF, dtype=np.float32)
    left[0] = gaussians[0].log_pdf(x[0]) + ltp.get(0, 0)
    right = np.full((N,), NEG_INF, dtype=np.float32)
    tracer = np.zeros((T, N), dtype=np.int8) - 1
    for t in range(1, T):
        for s in range(N):
            cand = np.full((N,), NEG_INF)                      # float64 scratch (line 180)
            for o in range(max(s - 2, 0), s + 1):
                cand[o] = +ltp.get(o, s) + left[o]             # float32 add
            best = np.max(cand)
            arg = np.argmax(cand)
            right[s] = best + gaussians[s].log_pdf(x[t])       # f64 + f32 -> stored f32
            tracer[t, s] = arg
        left = right
        right = np.full((N,), NEG_INF, dtype=np.float32)
    return left[-1], _backtrace(tracer, N - 1)


def viterbi_loop_ref(x: np.ndarray, ltp: LogTransRef, gaussians: Sequence[GaussianRef],
                     bounds: BoundariesRef, between) -> Tuple[np.float32, np.ndarray]:
    """Free word-loop decode (hidden_markov_model.py:463-479 init, 481-581).
    ``between`` keeps its Python/NumPy type: an int or Python float adds in
    float32, an ``np.float64`` (the class default np.log(0.005)) adds in float64."""
    N = len(gaussians)
    T = x.shape[0]
    W = bounds.num_words
    left = np.full((N,), NEG_INF, dtype=np.float32)
    for lb in bounds.lower:
        left[lb] = gaussians[lb].log_pdf(x[0]) + ltp.get(lb, lb)
    right = np.full((N,), NEG_INF, dtype=np.float32)
    tracer = np.zeros((T, N), dtype=np.int8) - 1
    for t in range(1, T):
        for s in range(N):
            if s in bounds.lower:
                continue
            lb = bounds.find_lower(s)
            cand = np.full((N,), NEG_INF)
            for o in range(max(s - 2, lb), s + 1):
                cand[o] = +ltp.get(o, s) + left[o]
            best = np.max(cand)
            arg = int(np.argmax(cand))
            right[s] = best + gaussians[s].log_pdf(x[t])
            tracer[t, s] = arg
        for s in bounds.lower:
            cand = np.full((W + 1,), NEG_INF)
            cand[-1] = +ltp.get(s, s) + left[s]
            for w, ub in enumerate(bounds.upper):
                cand[w] = +between + left[ub]
            best = np.max(cand)
            arg = int(np.argmax(cand))
            src = s if arg == W else bounds.upper[arg]
            right[s] = best + gaussians[s].log_pdf(x[t])
            tracer[t, s] = src
        left = right
        right = np.full((N,), NEG_INF, dtype=np.float32)
    ends = left[bounds.upper]
    best_score = np.max(ends)
    terminal = bounds.upper[int(np.argmax(ends))]
    return best_score, _backtrace(tracer, terminal)


# ----------------------------------------------------------------------------
# Viterbi, vectorised restatement on a precomputed emission matrix
# ----------------------------------------------------------------------------
def band_from_ltp(ltp: LogTransRef, n: int, lower_of_state: Optional[Sequence[int]] = None) -> np.ndarray:
    """float32 [3, n]: band[d, s] = log a(s-d -> s); -inf where the predecessor is
    not a candidate (s-d < 0, or below the word's first state when
    ``lower_of_state`` is given, as in the loop decoder's interior states)."""
    band = np.full((3, n), NEG_INF, dtype=np.float32)
    for s in range(n):
        floor = 0 if lower_of_state is None else lower_of_state[s]
        for d in range(3):
            o = s - d
            if o >= max(floor, 0):
                band[d, s] = np.float32(ltp.get(o, s))
    return band


def _band_step(prev: np.ndarray, band: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """One interior Viterbi step for all states: float32 candidate sums, first
    maximal predecessor wins, pointer 0 when every candidate is -inf
    (np.argmax over an all -inf array, hidden_markov_model.py:185-186)."""
    n = prev.shape[0]
    idx = np.arange(n)
    p1 = np.full(n, NEG_INF, dtype=np.float32)
    p2 = np.full(n, NEG_INF, dtype=np.float32)
    p1[1:] = prev[:-1]
    p2[2:] = prev[:-2]
    with np.errstate(invalid="ignore"):
        v0 = band[0] + prev
        v1 = band[1] + p1
        v2 = band[2] + p2
    best = np.maximum(np.maximum(v2, v1), v0)
    ptr = np.where(v2 == best, idx - 2, np.where(v1 == best, idx - 1, idx))
    ptr = np.where(best == NEG_INF, 0, ptr)
    return best, ptr


def viterbi_chain_fast(emis: np.ndarray, band: np.ndarray, start_mask: Optional[np.ndarray] = None,
                       terminal: Optional[int] = None) -> Tuple[np.float32, np.ndarray, np.ndarray]:
    """Banded recursion on float32 emissions [T, N]; returns (score, path, final
    likelihood vector).  ``start_mask`` marks the states initialised at frame 0
    (state 0 only for a single model)."""
    T, N = emis.shape
    if start_mask is None:
        start_mask = np.zeros(N, dtype=bool)
        start_mask[0] = True
    if terminal is None:
        terminal = N - 1
    prev = np.full(N, NEG_INF, dtype=np.float32)
    prev[start_mask] = emis[0, start_mask] + band[0, start_mask]
    tracer = np.zeros((T, N), dtype=np.int8) - 1
    for t in range(1, T):
        best, ptr = _band_step(prev, band)
        prev = (best.astype(np.float64) + emis[t].astype(np.float64)).astype(np.float32)
        tracer[t] = ptr.astype(np.int8)
    return prev[terminal], _backtrace(tracer, terminal), prev


def viterbi_loop_fast(emis: np.ndarray, band: np.ndarray, lower: Sequence[int], upper: Sequence[int],
                      between) -> Tuple[np.float32, np.ndarray]:
    """Vectorised loop decode.  ``band`` must already exclude predecessors below
    each word's first state (see ``band_from_ltp``)."""
    T, N = emis.shape
    lower = np.asarray(lower)
    upper = np.asarray(upper)
    wide = isinstance(between, np.float64)
    prev = np.full(N, NEG_INF, dtype=np.float32)
    prev[lower] = emis[0, lower] + band[0, lower]
    tracer = np.zeros((T, N), dtype=np.int8) - 1
    for t in range(1, T):
        best, ptr = _band_step(prev, band)
        best = best.astype(np.float64)
        if wide:
            cross = np.float64(between) + prev[upper].astype(np.float64)
        else:
            cross = (np.float32(between) + prev[upper]).astype(np.float64)
        m = np.max(cross)
        w = int(np.argmax(cross))
        self_loop = (band[0, lower] + prev[lower]).astype(np.float64)
        take_cross = m >= self_loop
        best[lower] = np.where(take_cross, m, self_loop)
        ptr[lower] = np.where(take_cross, upper[w], lower)
        prev = (best + emis[t].astype(np.float64)).astype(np.float32)
        tracer[t] = ptr.astype(np.int8)
    ends = prev[upper]
    terminal = int(upper[int(np.argmax(ends))])
    return np.max(ends), _backtrace(tracer, terminal)


# ----------------------------------------------------------------------------
# Hard-alignment statistics / M-step  (signal.py:23-47,68-91; hidden_markov_model.py:320-350)
# ----------------------------------------------------------------------------
def split_by_state(n_states: int, frames: np.ndarray, path: np.ndarray) -> List[Optional[np.ndarray]]:
    """signal.py:23-47: walk the states in ascending order, each takes the run of
    frames whose path value equals it starting where the previous state stopped."""
    out: List[Optional[np.ndarray]] = []
    start = 0
    for s in range(n_states):
        end = start
        for v in path[start:]:
            if v == s:
                end += 1
            else:
                break
        out.append(frames[start:end] if start < end else None)
        start = end
    return out


class EmptyStateError(Exception):
    """A state received no frames (hidden_markov_model.py:326-329, HMMTrainMeanFail)."""


class Converged(Exception):
    """New means are np.allclose to the old ones (hidden_markov_model.py:333-335)."""


def m_step(n_states: int, segments: Sequence[Tuple[np.ndarray, np.ndarray]], old_means: np.ndarray,
           old_cov: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """One re-estimation from (frames, path) pairs; returns (means f32, cov f32,
    transition probabilities f32 [N,N]).  Raises ``Converged`` before touching
    covariances/transitions, exactly where the reference does."""
    per_state: List[List[np.ndarray]] = [[] for _ in range(n_states)]
    for frames, path in segments:
        for s, part in enumerate(split_by_state(n_states, frames, path)):
            if part is not None:
                per_state[s].append(part)
    try:
        stacked = [np.concatenate(parts) for parts in per_state]
    except ValueError:
        raise EmptyStateError
    new_means = [np.average(a, axis=0) for a in stacked]
    if np.allclose(new_means, old_means):
        raise Converged
    means = np.array(new_means, dtype=np.float32)
    cov = np.array(old_cov, dtype=np.float32, copy=True)
    for s, a in enumerate(stacked):
        cov[s] = (np.cov(a, rowvar=False) + np.eye(cov.shape[1]) * 0.001).astype(np.float32)
    counts = np.zeros((n_states, n_states), dtype=np.int32)          # signal.py:80-91
    for _, path in segments:
        last = path[0]
        for cur in path[1:]:
            counts[last, cur] += 1
            last = cur
    with np.errstate(invalid="ignore", divide="ignore"):
        trans = (counts / np.sum(counts, axis=1, keepdims=True)).astype(np.float32)
    return means, cov, trans


def init_parameters(first_utt: np.ndarray, n_states: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """hidden_markov_model.py:359-389: uniform split of the FIRST utterance only."""
    D = first_utt.shape[1]
    seg = int(first_utt.shape[0] / n_states)
    means = [np.average(first_utt[i * seg: i * seg + seg, :], axis=0) for i in range(n_states)]
    cov = (np.tile(np.eye(D), (n_states, 1, 1)) * 0.01).astype(np.float32)
    return np.array(means, dtype=np.float32), cov, uniform_upper_triangular(n_states)


class WordModelRef:
    """Trainable left-to-right word model (hidden_markov_model.py:211-410)."""

    def __init__(self, label: str, means: np.ndarray, cov: np.ndarray, trans: np.ndarray):
        self.label = label
        self.means = np.asarray(means, dtype=np.float32)
        self.cov = np.asarray(cov, dtype=np.float32)
        self.trans = np.asarray(trans, dtype=np.float32)
        self.refresh()

    @property
    def n(self) -> int:
        return self.means.shape[0]

    def refresh(self) -> None:                                 # hidden_markov_model.py:283-292
        self.ltp = LogTransRef.from_probs(self.trans)
        self.gauss = [GaussianRef(m, c) for m, c in zip(self.means, self.cov)]

    def align(self, x: np.ndarray) -> Tuple[np.float32, np.ndarray]:
        return viterbi_chain_ref(x, self.ltp, self.gauss)

    def align_fast(self, x: np.ndarray) -> Tuple[np.float32, np.ndarray]:
        mu, U, ld = export_gaussians(self.gauss)
        score, path, _ = viterbi_chain_fast(emission_fast(x, mu, U, ld), band_from_ltp(self.ltp, self.n))
        return score, path

    def reestimate(self, segments: Sequence[Tuple[np.ndarray, np.ndarray]]) -> None:
        self.means, self.cov, self.trans = m_step(self.n, segments, self.means, self.cov)

    @classmethod
    def from_data(cls, label: str, utts: Sequence[np.ndarray], n_states: int = 5, max_iterations: int = 100,
                  fast: bool = True) -> "WordModelRef":
        """hidden_markov_model.py:233-281."""
        m = cls(label, *init_parameters(utts[0], n_states))
        for _ in range(max_iterations):
            try:
                segs = [(x, (m.align_fast(x) if fast else m.align(x))[1]) for x in utts]
                m.reestimate(segs)
                m.refresh()
            except Converged:
                break
        m.refresh()
        return m


def export_gaussians(gaussians: Sequence[GaussianRef]) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    mu, U, ld = zip(*(g.export() for g in gaussians))
    return np.stack(mu), np.stack(U), np.array(ld, dtype=np.float64)


def stack_models(models: Sequence[WordModelRef]) -> Tuple[LogTransRef, List[GaussianRef], BoundariesRef]:
    """Concatenate word models into one state space (hidden_markov_model.py:421-456, 638-664)."""
    ltp = LogTransRef()
    gauss: List[GaussianRef] = []
    for m in models:
        ltp.append(m.ltp)
        gauss.extend(m.gauss)
    return ltp, gauss, BoundariesRef([m.n for m in models], [m.label for m in models])


def insert_silence(labels: str) -> str:                        # hidden_markov_model.py:794-797
    return "".join(f"S{c}" for c in labels) + "S"


def remux(frames: np.ndarray, path: np.ndarray, bounds: BoundariesRef
          ) -> Dict[str, List[Tuple[int, np.ndarray, np.ndarray]]]:
    """hidden_markov_model.py:602-636: cut the aligned utterance wherever the word
    LABEL changes; each piece becomes (n_states, frames, path - word offset).
    The last piece is never appended (only a label change flushes)."""
    out: Dict[str, List[Tuple[int, np.ndarray, np.ndarray]]] = {lab: [] for lab in bounds.labels}
    last_i = 0
    last_state = path[0]
    last_label = bounds.label_of(int(last_state))
    for i, st in enumerate(path):
        lab = bounds.label_of(int(st))
        if lab != last_label:
            lo = bounds.find_lower(int(last_state))
            hi = bounds.find_upper(int(last_state))
            out[last_label].append((hi - lo + 1, frames[last_i:i], path[last_i:i] - lo))
            last_i = i
            last_state = path[i]
            last_label = bounds.label_of(int(last_state))
    return out


def continuous_iteration(models: Dict[str, WordModelRef], labeled: Dict[str, Sequence[np.ndarray]],
                         fast: bool = True) -> Tuple[Dict[str, List[Tuple[int, np.ndarray, np.ndarray]]], List[str]]:
    """One pass of hidden_markov_model.py:733-770: forced alignment of every
    utterance against S d S d ... S, re-mux per word, then the per-word M-step.
    Returns the re-muxed pieces and the labels whose M-step reported convergence."""
    pieces: Dict[str, List[Tuple[int, np.ndarray, np.ndarray]]] = {lab: [] for lab in models}
    for digits, utts in labeled.items():
        chain = insert_silence(digits)
        ltp, gauss, bounds = stack_models([models[c] for c in chain])
        if fast:
            mu, U, ld = export_gaussians(gauss)
            band = band_from_ltp(ltp, len(gauss))
        for x in utts:
            if fast:
                _, path, _ = viterbi_chain_fast(emission_fast(x, mu, U, ld), band)
            else:
                _, path = viterbi_chain_ref(x, ltp, gauss)
            for lab, segs in remux(x, path, bounds).items():
                pieces[lab].extend(segs)
    converged: List[str] = []
    for lab, segs in pieces.items():
        m = models[lab]
        try:
            m.reestimate([(f, p) for _, f, p in segs])
        except Converged:
            converged.append(lab)
        finally:
            m.refresh()
    return pieces, converged


def classify_isolated(models: Sequence[WordModelRef], x: np.ndarray, fast: bool = True) -> str:
    """model_collection.py:23-28: best single-model score; Python's stable sort with
    reverse=True keeps the earliest model among equal scores."""
    scores = {m.label: (m.align_fast(x) if fast else m.align(x))[0] for m in models}
    return [lab for lab, _ in sorted(scores.items(), key=lambda kv: kv[1], reverse=True)][0]
