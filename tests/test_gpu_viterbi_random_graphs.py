"""Loop decoding of randomly shaped word graphs against the CPU restatement (itself bit-identical to the reference
on its fixtures, tests/test_oracle_hmm.py), from the SAME emission scores: the device's score matrix is copied back and
fed to the restatement, so every difference would be a difference in the recursion, the tie rules, the back-trace or
the label pass.  Half of the graphs give all states one and the same Gaussian and draw transition rows from a few
patterns, so that ties (equal predecessors, cross-word entry against self loop, equal word exits, equal final states)
occur at almost every frame; graphs beyond 64 states take the two-wavefront kernel."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from helpers import trainable  # noqa: E402
from oracle import hmm_oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu

ROWS = [(0.5, 0.5, 0.0), (0.5, 0.25, 0.25), (0.25, 0.5, 0.25), (0.625, 0.375, 0.0), (0.25, 0.25, 0.5)]
BETWEENS = [-100, -3.0, np.log(0.005), 0, np.float32(-7.5)]
NAMES = "abcdefghijklmnopqrtuvwxyzABCDEFGHIJKLMNOPQRTUVWXYZ"         # no "S": that label marks the silence word


def _transitions(rng, n):
    t = np.zeros((n, n), dtype=np.float32)
    for s in range(n):
        row = ROWS[rng.integers(len(ROWS))]
        if s == n - 1:
            row = (1.0, 0.0, 0.0)
        elif s == n - 2:
            row = (row[0], 1.0 - row[0], 0.0)
        for d in range(3):
            if s + d < n:
                t[s, s + d] = row[d]
    return t


def _graph(rng, total_limit, tied):
    sizes = []
    while True:
        n = int(rng.integers(1, 9))
        if sum(sizes) + n > total_limit:
            break
        sizes.append(n)
        if len(sizes) >= 2 and rng.random() < 0.08:
            break
    labels = list(NAMES[:len(sizes)])
    if rng.random() < 0.5:
        labels[int(rng.integers(len(labels)))] = "S"
    a = rng.standard_normal((39, 39))
    shared = (rng.standard_normal(39).astype(np.float32), (a @ a.T / 39 + 0.05 * np.eye(39)).astype(np.float32))
    words = []
    for lab, n in zip(labels, sizes):
        means, covs = [], []
        for _ in range(n):
            if tied:
                m, c = shared
            else:
                a = rng.standard_normal((39, 39))
                m, c = rng.standard_normal(39).astype(np.float32), (a @ a.T / 39 + 0.05 * np.eye(39)).astype(np.float32)
            means.append(m)
            covs.append(c)
        words.append((lab, np.stack(means), np.stack(covs), _transitions(rng, n)))
    return words


@pytest.mark.parametrize("seed", range(16))
def test_random_loop_graph(seed):
    from loe_speech_recognition_amd import HiddenMarkovModelInference, _native

    rng = np.random.default_rng(1000 + seed)
    wide = seed % 4 == 3
    words = _graph(rng, 127 if wide else 64, tied=seed % 2 == 0)
    if wide:
        while sum(w[1].shape[0] for w in words) <= 64:          # make sure the graph needs the wide kernel
            words = _graph(rng, 127, tied=seed % 2 == 0)
    inf = HiddenMarkovModelInference.from_models([trainable(*w) for w in words])
    ltp, gauss, bounds = O.stack_models([O.WordModelRef(*w) for w in words])
    S = bounds.num_states
    band = O.band_from_ltp(ltp, S, [bounds.find_lower(s) for s in range(S)])
    xs = [(1.5 * rng.standard_normal((int(rng.integers(9, 70)), 39))).astype(np.float32) for _ in range(12)]
    eng = _native.get_engine()
    for between in BETWEENS:
        inf._log_transition_probability_between_words = between
        vit = inf.viterbi_batch(xs)
        strings = inf.predict_batch(xs)
        inf._bind()
        ll = eng.emission(eng.feature_batch(xs)).cpu().numpy()
        assert np.all(np.isneginf(ll[:, S:]))                   # padding columns
        t0 = 0
        for i, x in enumerate(xs):
            emis = ll[t0:t0 + x.shape[0], :S]
            t0 += x.shape[0]
            score, path = O.viterbi_loop_fast(emis, band, bounds.lower, bounds.upper, between)
            assert vit[i][0] == score or (np.isnan(vit[i][0]) and np.isnan(score)), (seed, between, i)
            assert np.array_equal(vit[i][1], path), (seed, between, i)
            assert strings[i] == "".join(bounds.get_labels(path)), (seed, between, i)


@pytest.mark.parametrize("seed", range(10))
def test_random_alignment_chain(seed):
    """Forced-alignment chains of random words (1..127 states, free transitions across word boundaries) from the
    device's own emission scores; chains beyond 64 states take the two-wavefront kernel."""
    from loe_speech_recognition_amd import HiddenMarkovModelMultiWord, _native

    rng = np.random.default_rng(2000 + seed)
    tied = seed % 2 == 1
    words = _graph(rng, 40, tied)
    by_label = {w[0]: w for w in words}
    limit = 127 if seed % 3 == 2 else 64
    chain, n_states = "", 0
    while True:
        lab = words[int(rng.integers(len(words)))][0]
        n = by_label[lab][1].shape[0]
        if n_states + n > limit or (chain and rng.random() < 0.05):
            break
        chain += lab
        n_states += n
    hmm = HiddenMarkovModelMultiWord.from_labels(chain, {lab: trainable(*w) for lab, w in by_label.items()})
    assert hmm.num_of_states == n_states
    ltp, _, _ = O.stack_models([O.WordModelRef(*by_label[lab]) for lab in chain])
    band = O.band_from_ltp(ltp, n_states)
    # long enough that the last state can be reached, plus two utterances that are too short for that
    xs = [(1.5 * rng.standard_normal((int(rng.integers(n_states // 2 + 9, n_states + 80)), 39))).astype(np.float32)
          for _ in range(10)]
    xs += [(1.5 * rng.standard_normal((9, 39))).astype(np.float32), (1.5 * rng.standard_normal((1, 39))).astype(np.float32)]
    out = hmm.predict_batch(xs)
    eng = _native.get_engine()
    token, arrays = hmm._device_model()
    eng.set_model(token, *arrays)
    ll = eng.emission(eng.feature_batch(xs)).cpu().numpy()
    t0 = 0
    for i, x in enumerate(xs):
        emis = ll[t0:t0 + x.shape[0], :n_states]
        t0 += x.shape[0]
        score, path, _ = O.viterbi_chain_fast(emis, band)
        assert out[i][0] == score or (np.isnan(out[i][0]) and np.isnan(score)), (seed, i, out[i][0], score)
        assert np.array_equal(out[i][1], path), (seed, i)


def test_every_length_across_the_kernel_boundaries():
    """Utterances of 1 .. 140 frames in one batch: the recursion's single-frame head ((T - 1) % 8 frames), the
    four-frame back-pointer groups and the 64-frame blocks of the back-trace all change shape with T."""
    from loe_speech_recognition_amd import HiddenMarkovModelInference, HiddenMarkovModelMultiWord, _native

    rng = np.random.default_rng(4242)
    words = _graph(rng, 60, tied=True)
    inf = HiddenMarkovModelInference.from_models([trainable(*w) for w in words])
    ltp, gauss, bounds = O.stack_models([O.WordModelRef(*w) for w in words])
    S = bounds.num_states
    band = O.band_from_ltp(ltp, S, [bounds.find_lower(s) for s in range(S)])
    xs = [(1.5 * rng.standard_normal((t, 39))).astype(np.float32) for t in range(1, 141)]
    eng = _native.get_engine()
    for between in (-3, np.float64(-7.5)):
        inf._log_transition_probability_between_words = between
        vit = inf.viterbi_batch(xs)
        strings = [None] + inf.predict_batch(xs[1:])              # a one-frame utterance has no labels (the reference raises)
        inf._bind()
        ll = eng.emission(eng.feature_batch(xs)).cpu().numpy()
        t0 = 0
        for i, x in enumerate(xs):
            score, path = O.viterbi_loop_fast(ll[t0:t0 + x.shape[0], :S], band, bounds.lower, bounds.upper, between)
            t0 += x.shape[0]
            assert vit[i][0] == score, (between, i)
            assert np.array_equal(vit[i][1], path), (between, i)
            if i:
                assert strings[i] == "".join(bounds.get_labels(path)), (between, i)
    # the same lengths through the chain form (a chain of the first words, at most 64 states)
    by_label = {w[0]: w for w in words}
    chain, n_states = "", 0
    for w in words:
        if n_states + w[1].shape[0] > 40:
            break
        chain += w[0]
        n_states += w[1].shape[0]
    hmm = HiddenMarkovModelMultiWord.from_labels(chain, {lab: trainable(*w) for lab, w in by_label.items()})
    cltp, _, _ = O.stack_models([O.WordModelRef(*by_label[lab]) for lab in chain])
    cband = O.band_from_ltp(cltp, n_states)
    out = hmm.predict_batch(xs)
    token, arrays = hmm._device_model()
    eng.set_model(token, *arrays)
    ll = eng.emission(eng.feature_batch(xs)).cpu().numpy()
    t0 = 0
    for i, x in enumerate(xs):
        score, path, _ = O.viterbi_chain_fast(ll[t0:t0 + x.shape[0], :n_states], cband)
        t0 += x.shape[0]
        assert out[i][0] == score or (np.isnan(out[i][0]) and np.isnan(score)), i
        assert np.array_equal(out[i][1], path), i


def test_thousands_of_very_short_utterances():
    """3 000 utterances of 2 .. 24 frames: nearly every back-pointer group is partly filled and every utterance
    starts a group of its own, the shape in which the pointer table is largest relative to the frames."""
    from loe_speech_recognition_amd import HiddenMarkovModelInference, _native

    rng = np.random.default_rng(515)
    words = _graph(rng, 50, tied=False)
    inf = HiddenMarkovModelInference.from_models([trainable(*w) for w in words])
    inf._log_transition_probability_between_words = -4
    ltp, gauss, bounds = O.stack_models([O.WordModelRef(*w) for w in words])
    S = bounds.num_states
    band = O.band_from_ltp(ltp, S, [bounds.find_lower(s) for s in range(S)])
    xs = [(1.5 * rng.standard_normal((int(t), 39))).astype(np.float32) for t in rng.integers(2, 25, size=3000)]
    vit = inf.viterbi_batch(xs)
    strings = inf.predict_batch(xs)
    eng = _native.get_engine()
    inf._bind()
    ll = eng.emission(eng.feature_batch(xs)).cpu().numpy()
    t0 = 0
    for i, x in enumerate(xs):
        score, path = O.viterbi_loop_fast(ll[t0:t0 + x.shape[0], :S], band, bounds.lower, bounds.upper, -4)
        t0 += x.shape[0]
        assert vit[i][0] == score and np.array_equal(vit[i][1], path), i
        assert strings[i] == "".join(bounds.get_labels(path)), i


def test_utterance_longer_than_64k_frames_and_the_limit():
    """One utterance of 70 000 frames (its path needs more than 64 KiB of LDS, a size the launcher has to ask for)
    next to a short one; beyond the LDS of a compute unit the decoder refuses with a clear error."""
    from loe_speech_recognition_amd import HiddenMarkovModelInference, _native

    rng = np.random.default_rng(77)
    words = _graph(rng, 40, tied=False)
    inf = HiddenMarkovModelInference.from_models([trainable(*w) for w in words])
    inf._log_transition_probability_between_words = -20
    ltp, gauss, bounds = O.stack_models([O.WordModelRef(*w) for w in words])
    S = bounds.num_states
    band = O.band_from_ltp(ltp, S, [bounds.find_lower(s) for s in range(S)])
    xs = [(1.5 * rng.standard_normal((70000, 39))).astype(np.float32), (1.5 * rng.standard_normal((30, 39))).astype(np.float32)]
    vit = inf.viterbi_batch(xs)
    strings = inf.predict_batch(xs, label_cap=32)
    eng = _native.get_engine()
    inf._bind()
    ll = eng.emission(eng.feature_batch(xs)).cpu().numpy()
    t0 = 0
    for i, x in enumerate(xs):
        score, path = O.viterbi_loop_fast(ll[t0:t0 + x.shape[0], :S], band, bounds.lower, bounds.upper, -20)
        t0 += x.shape[0]
        assert vit[i][0] == score and np.array_equal(vit[i][1], path), i
        assert strings[i] == "".join(bounds.get_labels(path)), i            # the long one overflows its record
    with pytest.raises(_native.NativeError, match="frame"):
        inf.viterbi_batch([np.zeros((170000, 39), np.float32)])
