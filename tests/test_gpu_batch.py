"""Batched decode on the device against the oracle on larger seeded batches, plus
size-independent properties at the benchmark's full batch size."""
import numpy as np
import pytest

from oracle import hmm_oracle as O
from oracle import mfcc_oracle as M

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def decoder():
    from loe_speech_recognition_amd import HiddenMarkovModelInference, synthetic

    models = synthetic.load_word_models()
    inf = HiddenMarkovModelInference.from_models([models[lab] for lab in synthetic.LOOP_ORDER])
    inf._log_transition_probability_between_words = -100
    return inf


@pytest.fixture(scope="module")
def oracle_graph():
    from loe_speech_recognition_amd import synthetic

    arrays = synthetic.load_word_arrays()
    words = [O.WordModelRef(lab, *arrays[lab]) for lab in synthetic.LOOP_ORDER]
    ltp, gauss, bounds = O.stack_models(words)
    mu, U, ld = O.export_gaussians(gauss)
    band = O.band_from_ltp(ltp, bounds.num_states, [bounds.find_lower(s) for s in range(bounds.num_states)])
    return mu, U, ld, band, bounds


def oracle_decode(graph, x, between=-100):
    mu, U, ld, band, bounds = graph
    score, path = O.viterbi_loop_fast(O.emission_fast(x, mu, U, ld), band, bounds.lower, bounds.upper, between)
    return score, path, "".join(bounds.get_labels(path))


@pytest.mark.parametrize("config,count,digits", [(2, 96, 1), (3, 128, 4), (5, 48, 7)])
def test_strings_equal_cpu_path(decoder, oracle_graph, config, count, digits):
    """Configs 2/3/5 at reduced count: audio -> strings on the device equals the oracle run on
    the oracle's own MFCCs (end-to-end) and on the device's MFCCs (decode stage alone)."""
    from loe_speech_recognition_amd import MFCC, synthetic

    sigs, truth = synthetic.batch(config, count, digits)
    fused = decoder.recognize_audio(sigs)
    feats = MFCC.batch(sigs, 16000)
    staged = decoder.predict_batch(feats)
    vit = decoder.viterbi_batch(feats)
    assert fused == staged
    mismatch_e2e = 0
    for i, s in enumerate(sigs):
        score, path, lab = oracle_decode(oracle_graph, feats[i])
        assert staged[i] == lab
        assert np.array_equal(vit[i][1], path) and vit[i][0] == score
        mismatch_e2e += oracle_decode(oracle_graph, M.mfcc_features(s).T)[2] != fused[i]
    assert mismatch_e2e == 0
    assert np.mean([a == b for a, b in zip(fused, truth)]) >= 0.9


def test_no_silence_graph_between_250(oracle_graph):
    """Config 5's second variant: 11 digit models without the silence word, between = -250
    (scripts/project5_test_ndigits_no_sil.py:59)."""
    from loe_speech_recognition_amd import MFCC, HiddenMarkovModelInference, synthetic

    order = [lab for lab in synthetic.LOOP_ORDER if lab != "S"]
    models = synthetic.load_word_models()
    inf = HiddenMarkovModelInference.from_models([models[lab] for lab in order])
    inf._log_transition_probability_between_words = -250
    arrays = synthetic.load_word_arrays()
    ltp, gauss, bounds = O.stack_models([O.WordModelRef(lab, *arrays[lab]) for lab in order])
    mu, U, ld = O.export_gaussians(gauss)
    band = O.band_from_ltp(ltp, 55, [bounds.find_lower(s) for s in range(55)])
    sigs, _ = synthetic.batch(5, 32, 7, start=500)
    feats = MFCC.batch(sigs, 16000)
    got = inf.viterbi_batch(feats)
    labels = inf.predict_batch(feats)
    for i in range(len(sigs)):
        score, path = O.viterbi_loop_fast(O.emission_fast(feats[i], mu, U, ld), band, bounds.lower, bounds.upper, -250)
        assert got[i][0] == score and np.array_equal(got[i][1], path)
        assert labels[i] == "".join(bounds.get_labels(path))


def test_short_and_degenerate_sequences(decoder, oracle_graph):
    rng = np.random.default_rng(11)
    xs = [rng.standard_normal((t, 39)).astype(np.float32) * 3 for t in (2, 3, 5, 63, 64, 65, 128, 129, 700)]
    got = decoder.viterbi_batch(xs)
    labels = decoder.predict_batch(xs)
    for x, (score, path), lab in zip(xs, got, labels):
        s, p, l = oracle_decode(oracle_graph, x)
        assert score == s and np.array_equal(path, p) and lab == l
    # a one-frame utterance has no back-pointer row: the reference returns path [-1]
    score, path = decoder.viterbi_batch([xs[0][:1]])[0]
    assert path.tolist() == [-1]
    with pytest.raises(Exception):
        decoder.predict(xs[0][:1])


def test_label_record_overflow_falls_back_to_path(decoder, oracle_graph):
    from loe_speech_recognition_amd import MFCC, synthetic

    sigs, _ = synthetic.batch(5, 4, 7, start=900)
    feats = MFCC.batch(sigs, 16000)
    assert decoder.predict_batch(feats, label_cap=2) == decoder.predict_batch(feats, label_cap=32)


def test_full_batch_properties(decoder):
    """10 000 four-digit utterances (the benchmark workload size): 2 000 distinct clips, each
    present five times at shuffled positions.  Checks that do not need the slow CPU path:
    duplicates decode identically wherever they sit in the batch, device label records equal
    the host rule applied to the device paths, paths respect the graph, and the strings match
    the generating digit strings."""
    import multiprocessing as mp

    from loe_speech_recognition_amd import _native, synthetic

    with mp.get_context("fork").Pool(8) as pool:
        parts = pool.starmap(synthetic.batch, [(3, 250, 4, 250 * k) for k in range(8)])
    sigs = [s for p in parts for s in p[0]]
    truth = [d for p in parts for d in p[1]]
    rng = np.random.default_rng(3)
    order = rng.permutation(np.tile(np.arange(2000), 5))
    eng = decoder._bind()
    ab = eng.audio_batch([sigs[i] for i in order])
    assert ab.n == 10000
    fb = eng.mfcc(ab)
    scores, paths, labels, n_labels = eng.decode_loop(fb, want_paths=True)
    strings = decoder._labels_from_device(eng, fb, labels, n_labels, 32)
    scores = scores.cpu().numpy()
    plist = _native.split_ragged(paths.cpu().numpy(), fb.host_off)
    first = {}
    mb = decoder._model_boundaries
    lower = np.array(mb.lower_boundaries)
    for pos, src in enumerate(order):
        if src in first:
            q = first[src]
            assert strings[pos] == strings[q] and scores[pos] == scores[q] and np.array_equal(plist[pos], plist[q])
        else:
            first[src] = pos
            p = plist[pos].astype(np.int64)
            assert "".join(mb.get_labels(p)) == strings[pos]
            step = np.diff(p)
            ok = (step >= 0) & (step <= 2) | np.isin(p[1:], lower)      # stay/advance inside a word or enter a word
            assert ok.all() and p[0] in lower
    acc = np.mean([strings[pos] == truth[src] for pos, src in enumerate(order)])
    assert acc >= 0.95, acc


def test_penalty_sweep_equals_one_decode_per_penalty(decoder, oracle_graph):
    """predict_sweep (one emission pass, one decode per penalty) returns what re-decoding after each
    assignment of the between-word penalty returns, for int, float and np.float64 penalties, and the
    penalty left on the object is untouched."""
    from loe_speech_recognition_amd import MFCC, synthetic

    sigs, _ = synthetic.batch(3, 40, 2)
    feats = MFCC.batch(sigs, 16000)
    betweens = [0, -50, -100, -250.5, np.log(0.005), np.float64(-30.0), -2000, np.float32(-75.0), -1e5, 40.0]
    swept = decoder.predict_sweep(feats, betweens)
    assert len(swept) == len(betweens)
    saved = decoder._log_transition_probability_between_words
    try:
        for b, row in zip(betweens, swept):
            decoder._log_transition_probability_between_words = b
            assert decoder.predict_batch(feats) == row
            for i in (0, 7, 21):
                assert oracle_decode(oracle_graph, feats[i], b)[2] == row[i]
    finally:
        decoder._log_transition_probability_between_words = saved
    assert decoder.predict_batch(feats) == swept[2]
    assert len({tuple(r) for r in swept}) > 1          # the penalty matters on this data
    with pytest.raises(ValueError):
        decoder.predict_sweep(feats, [])
