"""CPU-only checks: the C-ABI library loads and exports every declared symbol, the host-side
tables match the oracle, and the host bookkeeping classes follow the reference's rules."""
import os
import re

import numpy as np
import pytest

from helpers import LOOP_LABELS, load
from oracle import hmm_oracle as O
from oracle import mfcc_oracle as M

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from loe_speech_recognition_amd import _native

    lib = _native.load_library()
    header = open(os.path.join(ROOT, "include", "lsr_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(lsr_[a-z0-9_]+)\s*\(", header)))
    assert declared and sorted(_native.EXPORTED_SYMBOLS) == declared
    for name in declared:
        assert getattr(lib, name) is not None


def test_host_tables_match_oracle():
    import scipy.signal

    from loe_speech_recognition_amd import _native

    win, mel, dct = _native.host_tables()
    assert np.array_equal(win, scipy.signal.get_window("hann", 320, fftbins=True))
    ref = M.mel_filterbank()
    assert np.all(np.abs(mel - ref) <= np.spacing(ref)) and np.mean(mel != ref) < 1e-3
    n = np.arange(40)
    k = np.arange(13)[:, None]
    basis = np.cos(np.pi * k * (2 * n + 1) / 80.0) * np.sqrt(2.0 / 40)
    basis[0] *= np.sqrt(0.5)
    assert np.allclose(dct, basis, atol=1e-15)


def test_engine_needs_gpu():
    import torch

    from loe_speech_recognition_amd import _native

    if not torch.cuda.is_available():
        with pytest.raises(_native.NativeError):
            _native.Engine()


def test_engine_refuses_a_child_forked_after_gpu_use(monkeypatch):
    """HIP does not survive fork(): an engine requested in a process other than the one that first initialised the GPU
    (which is what a forked child looks like: it inherits the parent's module state) raises instead of faulting."""
    import os

    from loe_speech_recognition_amd import _native

    monkeypatch.setattr(_native, "_hip_pid", os.getpid() + 1)
    monkeypatch.setattr(_native, "_engine", None)
    with pytest.raises(_native.NativeError, match="spawn"):
        _native.get_engine()


def test_model_boundary_labels_match_reference():
    from loe_speech_recognition_amd import ModelBoundary

    g = load("hmm_g4_labels.npz")
    mb = ModelBoundary()
    for lab in LOOP_LABELS:
        mb.append(3 if lab == "S" else 5)
    mb.add_model_labels(list(LOOP_LABELS))
    assert mb.lower_boundaries == [0, 5, 10, 15, 20, 25, 30, 35, 40, 45, 50, 53]
    assert mb.upper_boundaries == [4, 9, 14, 19, 24, 29, 34, 39, 44, 49, 52, 57]
    for i in range(int(g["n"])):
        assert "".join(mb.get_labels(g[f"p{i}"])) == str(g[f"l{i}"])
        assert "".join(mb.get_labels(g[f"p{i}"], skip_silence=False)) == str(g[f"k{i}"])
    with pytest.raises(Exception):
        mb.append(5)
    assert mb.get_state_range("2") == (5, 10)


def test_transition_store_semantics():
    from loe_speech_recognition_amd import LogTransitionProbabilities, TransitionProbabilities

    tp = TransitionProbabilities.from_num_of_states(3)
    assert tp[(0, 0)] == np.float32(1 / 3) and tp[(1, 0)] == 0 and tp[(2, 2)] == 1
    ltp = LogTransitionProbabilities.from_transition_probability(tp)
    assert ltp[(1, 0)] == -np.inf and ltp[(0, 1)] == np.log(np.float32(1 / 3))
    both = LogTransitionProbabilities()
    both.append(ltp)
    both.append(ltp)
    assert both.num_of_states == 6 and both[(2, 3)] == 0.0 and both[(3, 4)] == ltp[(0, 1)]
    band = both.band()
    ref = O.LogTransRef.from_probs(tp.to_array())
    stacked = O.LogTransRef()
    stacked.append(ref)
    stacked.append(ref)
    assert np.array_equal(band, O.band_from_ltp(stacked, 6))
    assert band[1, 3] == 0.0 and band[2, 3] == 0.0 and np.isneginf(band[1, 0])
    floor = both.band([0, 0, 0, 3, 3, 3])
    assert np.isneginf(floor[1, 3]) and np.isneginf(floor[2, 4]) and floor[1, 4] == band[1, 4]


def test_signal_grouping_and_remux_match_oracle():
    from loe_speech_recognition_amd import HiddenMarkovModelMultiWord, ModelBoundary, Signal, SortedSignals

    rng = np.random.default_rng(0)
    frames = rng.standard_normal((9, 39)).astype(np.float32)
    path = np.array([0, 0, 2, 2, 3, 3, 3, 1, 4], dtype=np.int8)
    parts = Signal(5, frames, path).order_by_state
    ref = O.split_by_state(5, frames, path)
    for a, b in zip(parts, ref):
        assert (a is None and b is None) or np.array_equal(a, b)
    ss = SortedSignals(5)
    ss.append(Signal(5, frames, np.array([0, 0, 1, 1, 2, 3, 3, 4, 4], dtype=np.int8)))
    assert ss.transition_counts[0, 0] == 1 and ss.transition_counts[3, 4] == 1
    mb = ModelBoundary()
    for n in (3, 5, 3):
        mb.append(n)
    mb.add_model_labels(["S", "1", "S"])
    p = np.array([0, 1, 2, 3, 4, 6, 7, 8, 9], dtype=np.int8)
    got = HiddenMarkovModelMultiWord._remux_path_and_signal(frames, p, mb)
    exp = O.remux(frames, p, O.BoundariesRef([3, 5, 3], ["S", "1", "S"]))
    assert {k: len(v) for k, v in got.items()} == {k: len(v) for k, v in exp.items()} == {"S": 1, "1": 1}
    assert np.array_equal(got["1"][0].path, exp["1"][0][2]) and got["1"][0].num_of_state == 5


def test_insert_silence():
    from loe_speech_recognition_amd import HiddenMarkovModelTrainContinuous

    assert HiddenMarkovModelTrainContinuous.insert_silence("4Z2") == "S4SZS2S" == O.insert_silence("4Z2")


def test_mfcc_input_validation_needs_no_gpu():
    from loe_speech_recognition_amd import MFCC

    with pytest.raises(TypeError):
        MFCC([0.0] * 2000, 16000)
    with pytest.raises(ValueError):
        MFCC(np.zeros((2, 2000), np.float32), 16000)
    with pytest.raises(ValueError):
        MFCC(np.zeros(2000, np.float32), 8000)


def test_synthetic_models_load_and_are_positive_definite():
    from loe_speech_recognition_amd import synthetic

    arrays = synthetic.load_word_arrays()
    assert sorted(arrays) == sorted(synthetic.WORDS + ["S"])
    for lab, (means, cov, trans) in arrays.items():
        n = 3 if lab == "S" else 5
        assert means.shape == (n, 39) and cov.shape == (n, 39, 39) and trans.shape == (n, n)
        for c in cov:
            assert np.linalg.eigvalsh(c.astype(np.float64)).min() > 0
    sig, truth = synthetic.utterance(3, 0, 4)
    sig2, truth2 = synthetic.utterance(3, 0, 4)
    assert truth == truth2 and np.array_equal(sig, sig2) and sig.dtype == np.float32


def test_records_to_strings_vectorised_path_equals_the_loop():
    """Fixed-width label records -> strings: the one-character fast path, multi-character names, counts beyond the
    record width, the -1 sentinel and padding bytes."""
    from loe_speech_recognition_amd import _native

    rng = np.random.default_rng(0)
    names = list("123456789OSZ")
    lab = rng.integers(0, 12, size=(500, 16)).astype(np.uint8)
    cnt = rng.integers(-1, 20, size=500).astype(np.int32)
    lab[5, 3:] = 0xFF
    cnt[5] = 3
    want = ["".join(names[w] for w in lab[i, :max(0, min(int(cnt[i]), 16))]) for i in range(500)]
    assert _native.records_to_strings(lab, cnt, names) == want
    long_names = ["one", "two"] + names[2:]
    want = ["".join(long_names[w] for w in lab[i, :max(0, min(int(cnt[i]), 16))]) for i in range(500)]
    assert _native.records_to_strings(lab, cnt, long_names) == want
    assert _native.records_to_strings(np.zeros((0, 4), np.uint8), np.zeros(0, np.int32), names) == []


def test_batched_factorisation_carries_the_bits_of_the_serial_path():
    """factor_covariances (threads inside the helper extension, SciPy's own dsyevr / dgeqrf) against
    MultivariateNormal.from_means_covariances state by state and numpy.linalg.qr: same eigen-whitening, log
    determinant and triangular factor to the bit, for any thread count and for a dealt-out subset; same
    exceptions for matrices SciPy refuses."""
    import numpy as np
    import pytest

    from loe_speech_recognition_amd import MultivariateNormal
    from loe_speech_recognition_amd.hidden_markov_model import factor_covariances
    from loe_speech_recognition_amd import _lsr_pyhelper as H
    import scipy.linalg.cython_lapack as lapack

    rng = np.random.default_rng(21)
    S, d = 23, 39
    a = rng.standard_normal((S, d, d)) * rng.uniform(0.1, 30, size=(S, 1, 1))
    cov = (a @ a.transpose(0, 2, 1) / d + 1e-3 * np.eye(d)).astype(np.float32)
    cov[5] = np.eye(d, dtype=np.float32) * 0.01                      # the trainers' initial covariance
    mu = rng.standard_normal((S, d)).astype(np.float32)
    serial = [MultivariateNormal.from_means_covariances(m, c) for m, c in zip(mu, cov)]
    tri = np.tril(np.swapaxes(np.linalg.qr(np.swapaxes(np.stack([g.whiten for g in serial]), 1, 2), mode="r"), 1, 2))
    for threads in (1, 3, 16):
        got = factor_covariances(mu, cov, threads=threads)
        for s in range(S):
            assert np.array_equal(got[s].whiten, serial[s].whiten) and got[s].log_pdet == serial[s].log_pdet, (threads, s)
            assert np.array_equal(got[s].mean, serial[s].mean) and np.array_equal(got[s].tri, tri[s]), (threads, s)
    # a dealt-out share touches only its own rows
    lam, U, L, st = np.zeros((S, d)), np.zeros((S, d, d)), np.zeros((S, d, d)), np.full(S, 9, np.int32)
    caps = lapack.__pyx_capi__
    H.factor_batch(caps["dsyevr"], caps["dgeqrf"], np.ascontiguousarray(cov, dtype=np.float64), lam, U, L, st, S, d, 2, 3, 4)
    assert np.array_equal(U[2::3], np.stack([g.whiten for g in serial])[2::3]) and (st[2::3] == 0).all()
    mask = np.ones(S, bool)
    mask[2::3] = False
    assert (U[mask] == 0).all() and (st[mask] == 9).all()
    # matrices SciPy refuses: the first failing state decides, with SciPy's exception types
    bad = cov.copy()
    bad[7] = 0
    with pytest.raises(np.linalg.LinAlgError):
        factor_covariances(mu, bad)
    bad[4] = -np.eye(d, dtype=np.float32)
    with pytest.raises(ValueError, match="positive semidefinite"):
        factor_covariances(mu, bad)
    bad[1, 0, 0] = np.nan
    with pytest.raises(ValueError, match="infs or NaNs"):
        factor_covariances(mu, bad)


def test_fused_m_steps_equal_the_per_model_form():
    """HiddenMarkovModelTrainContinuous._m_steps does the arithmetic of all word models in one pass; model by model
    (HiddenMarkovModelTrainable._m_step + _update_inference_weights, the reference's loop, hidden_markov_model.py:754-770)
    must give the same bits, including a model that reports convergence and keeps its parameters."""
    import numpy as np

    from test_distributed_cpu import _continuous_trainer, _word_stats
    from loe_speech_recognition_amd import HiddenMarkovModelTrainable

    sizes, count, sum_x, sum_xx, trans = _word_stats(31)
    fused, loop = _continuous_trainer(sizes), _continuous_trainer(sizes)
    # make model "B" converge: its current means already equal the new ones
    lo = sizes[0]
    for tc in (fused, loop):
        tc._trainable_models["B"]._means = (sum_x[lo:lo + sizes[1]] / count[lo:lo + sizes[1], None]).astype(np.float32)
    base, _ = fused._layout()
    fused._m_steps(base, count, sum_x, sum_xx, trans)
    for lab in loop._models_loaded:
        m = loop._trainable_models[lab]
        a, b = base[lab], base[lab] + m.num_of_states
        try:
            m._m_step(count[a:b], sum_x[a:b], sum_xx[a:b], trans[a:b, a:b])
        except HiddenMarkovModelTrainable.HMMTrainConverge:
            loop._num_of_finished_models += 1
        finally:
            m._update_inference_weights()
    assert fused._num_of_finished_models == loop._num_of_finished_models == 1
    for lab in loop._models_loaded:
        f, l = fused._trainable_models[lab], loop._trainable_models[lab]
        assert np.array_equal(f._means, l._means) and np.array_equal(f._covariances, l._covariances), lab
        assert np.array_equal(f._transition_probs.to_array(), l._transition_probs.to_array(), equal_nan=True), lab
        assert np.array_equal(f._log_transition_probs.to_array(), l._log_transition_probs.to_array(), equal_nan=True), lab
        assert np.array_equal(f._log_transition_probs.band(), l._log_transition_probs.band(), equal_nan=True), lab
        for g, h in zip(f._multivariate_normals, l._multivariate_normals):
            assert np.array_equal(g.whiten, h.whiten) and g.log_pdet == h.log_pdet and np.array_equal(g.mean, h.mean), lab
    # a state without frames stops at its model; the models before it are updated, those after it are not
    bad = count.copy()
    bad[lo] = 0
    tc = _continuous_trainer(sizes)
    before_c = tc._trainable_models["C"]._means.copy()
    import pytest
    with pytest.raises(HiddenMarkovModelTrainable.HMMTrainMeanFail):
        tc._m_steps(base, bad, sum_x, sum_xx, trans)
    assert np.array_equal(tc._trainable_models["A"]._means, fused._trainable_models["A"]._means)
    assert np.array_equal(tc._trainable_models["C"]._means, before_c)
