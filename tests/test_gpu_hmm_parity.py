"""HIP path (through the C ABI, behind the drop-in classes) against fixtures computed by the
real reference code.  Scores, paths and label strings must be identical; emission scores are
float64 inside and may differ from SciPy's BLAS summation order by one float32 ulp on rare
elements (tolerance of the task: 1e-4 relative)."""
import numpy as np
import pytest

from helpers import BETWEEN, LOOP_LABELS, load, loop_decoder, rel_err, trainable, trans_close, word

pytestmark = pytest.mark.gpu


def test_g1_emission():
    from loe_speech_recognition_amd import HiddenMarkovModelTrainable, _native
    from loe_speech_recognition_amd.hidden_markov_model import stack_gaussians

    g = load("hmm_g1_emission.npz")
    mvn = HiddenMarkovModelTrainable.get_multivariate_normals(g["means"], g["cov"])
    assert np.array_equal(np.stack([m.whiten for m in mvn[:4]]), g["U_first4"])
    assert np.array_equal(np.array([m.log_pdet for m in mvn]), g["logdet"])
    eng = _native.get_engine()
    for triangular in (False, True):          # SciPy's U as is / rotated to lower triangular form
        eng.set_model(None, *stack_gaussians(mvn), triangular=triangular)
        ll = eng.emission(eng.feature_batch([g["x"]])).cpu().numpy()
        got, ref = ll[:64, :58], g["ll"]
        assert rel_err(got, ref) <= 1e-4
        assert np.all(np.abs(got - ref) <= np.spacing(np.abs(ref)))
        assert np.mean(got != ref) < 2e-3
        assert np.all(np.isneginf(ll[:64, 58:]))
    # single-vector interface (hidden_markov_model.py:46-48)
    v = mvn[7].log_pdf(g["x"][3])
    assert v.dtype == np.float32 and abs(v - ref[3, 7]) <= np.spacing(abs(ref[3, 7]))


@pytest.mark.parametrize("tag", ["a", "b"])
def test_g9_mixture_emission(tag):
    """Mixture emissions against values the reference's own HMM_GMM.log_emission_prob returned
    (deprecated/gaussian_mixture_model.py:159-164; tests/golden/gmm_g9_emission.npz), rounded to float32 as the
    live code rounds its emissions: at most one float32 step apart, on few elements."""
    from loe_speech_recognition_amd import GaussianMixture, _native
    from loe_speech_recognition_amd.hidden_markov_model import stack_gaussians

    g = load("gmm_g9_emission.npz")
    mixtures = [GaussianMixture.from_weights_means_covariances(w, m, c)
                for w, m, c in zip(g[f"{tag}_weights"], g[f"{tag}_means"], g[f"{tag}_cov"])]
    eng = _native.get_engine()
    eng.set_model(None, *stack_gaussians(mixtures))
    x = g[f"{tag}_x"]
    want = g[f"{tag}_logprob"].astype(np.float32)
    got = eng.emission(eng.feature_batch([x])).cpu().numpy()[:x.shape[0], :want.shape[1]]
    assert rel_err(got, want) <= 1e-4
    assert np.all(np.abs(got - want) <= np.spacing(np.abs(want)))
    assert np.mean(got != want) < 0.02
    v = mixtures[1].log_pdf(x[3])
    assert v.dtype == np.float32 and abs(v - want[3, 1]) <= np.spacing(abs(want[3, 1]))


@pytest.mark.parametrize("n", [3, 5])
def test_g2_single_model(n):
    g = load("hmm_g2_single.npz")
    m = trainable("w", g[f"n{n}_means"], g[f"n{n}_cov"], g[f"n{n}_trans"])
    tags = ("T9", "T50", "T200", "late")
    out = m.predict_batch([g[f"n{n}_{t}_x"] for t in tags])
    for tag, (score, path) in zip(tags, out):
        assert score == g[f"n{n}_{tag}_score"], tag
        assert path.dtype == np.int8 and np.array_equal(path, g[f"n{n}_{tag}_path"]), tag
    score, path = m.predict(g[f"n{n}_T50_x"])
    assert isinstance(score, np.float32) and score == g[f"n{n}_T50_score"]


@pytest.mark.parametrize("tag", ["n35", "n59"])
def test_g2_alignment_chain(tag):
    from loe_speech_recognition_amd import HiddenMarkovModelMultiWord

    g = load("hmm_g2_single.npz")
    words = {lab: word(g, f"chain_{lab}", lab) for lab in "123S"}
    hmm = HiddenMarkovModelMultiWord.from_labels(str(g[f"{tag}_chain"]), words)
    assert hmm.num_of_states == int(tag[1:])
    for T in (50, 200):
        if f"{tag}_T{T}_x" not in g.files:
            continue
        score, path = hmm.predict(g[f"{tag}_T{T}_x"])
        assert score == g[f"{tag}_T{T}_score"]
        assert np.array_equal(path, g[f"{tag}_T{T}_path"])


@pytest.mark.parametrize("tag", ["n75", "n123"])
def test_g7_long_alignment_chain(tag):
    """Forced-alignment chains of 75 and 123 states (two wavefronts per utterance) against the reference's output."""
    from loe_speech_recognition_amd import HiddenMarkovModelMultiWord

    g = load("hmm_g7_wide.npz")
    words = {lab: word(g, f"chain_{lab}", lab) for lab in "123S"}
    hmm = HiddenMarkovModelMultiWord.from_labels(str(g[f"{tag}_chain"]), words)
    assert hmm.num_of_states == int(tag[1:])
    xs = [g[f"{tag}_T{T}_x"] for T in (160, 300)]
    for (score, path), T in zip(hmm.predict_batch(xs), (160, 300)):          # both in one batch
        assert score == g[f"{tag}_T{T}_score"]
        assert path.dtype == np.int8 and np.array_equal(path, g[f"{tag}_T{T}_path"])
    score, path = hmm.predict(xs[0][:1])                                     # one frame: no back-pointer
    assert path.tolist() == [-1]


def test_g7_loop_of_123_states():
    """Free word loop over 24 five-state words and silence (123 states, emission rows of 128 floats) against the
    reference: scores, paths, label strings, for int, np.float64 and float penalties."""
    from loe_speech_recognition_amd import HiddenMarkovModelInference

    g = load("hmm_g7_wide.npz")
    order = [str(x) for x in g["loop_order"]]
    inf = HiddenMarkovModelInference.from_models([word(g, f"loop_{lab}", lab) for lab in order])
    assert inf._model_boundaries.upper_boundaries[-1] == 122
    n = int(g["n_utts"])
    xs = [g[f"u{i}_x"] for i in range(n)]
    for tag in ("m100", "dflt", "m3f"):
        inf._log_transition_probability_between_words = BETWEEN[tag]
        labels = inf.predict_batch(xs)
        vit = inf.viterbi_batch(xs)
        for i in range(n):
            assert vit[i][0] == g[f"u{i}_{tag}_score"], (i, tag)
            assert np.array_equal(vit[i][1], g[f"u{i}_{tag}_path"]), (i, tag)
            assert labels[i] == str(g[f"u{i}_{tag}_labels"]), (i, tag)
    # the narrow graph still decodes after the wide one used the context (tables and strides are re-uploaded)
    g3 = load("hmm_g3_loop.npz")
    small = loop_decoder(g3)
    small._log_transition_probability_between_words = -100
    assert small.predict(g3["u4_x"]) == str(g3["u4_m100_labels"])


def test_g3_loop_decode():
    g = load("hmm_g3_loop.npz")
    inf = loop_decoder(g)
    n = int(g["n_utts"])
    xs = [g[f"u{i}_x"] for i in range(n)]
    for tag, between in BETWEEN.items():
        inf._log_transition_probability_between_words = between
        labels = inf.predict_batch(xs)
        vit = inf.viterbi_batch(xs)
        for i in range(n):
            assert vit[i][0] == g[f"u{i}_{tag}_score"], (i, tag)
            assert np.array_equal(vit[i][1], g[f"u{i}_{tag}_path"]), (i, tag)
            assert labels[i] == str(g[f"u{i}_{tag}_labels"]), (i, tag)
    inf._log_transition_probability_between_words = -100
    assert inf.predict(xs[4]) == str(g["u4_m100_labels"])


def test_g3_from_models_saved_by_the_reference():
    """tests/golden/models_g3 holds the twelve G3 word models as the reference's own save() wrote them, converted
    by tools/convert_reference_models.py: loaded with from_folder they must reproduce the reference's scores, paths
    and label strings exactly."""
    import os

    from helpers import GOLDEN
    from loe_speech_recognition_amd import HiddenMarkovModelInference

    g = load("hmm_g3_loop.npz")
    inf = HiddenMarkovModelInference.from_folder(os.path.join(GOLDEN, "models_g3"), list(LOOP_LABELS))
    assert inf._model_boundaries._labels == list(LOOP_LABELS)
    n = int(g["n_utts"])
    xs = [g[f"u{i}_x"] for i in range(n)]
    for tag, between in BETWEEN.items():
        inf._log_transition_probability_between_words = between
        labels = inf.predict_batch(xs)
        vit = inf.viterbi_batch(xs)
        for i in range(n):
            assert vit[i][0] == g[f"u{i}_{tag}_score"], (i, tag)
            assert np.array_equal(vit[i][1], g[f"u{i}_{tag}_path"]), (i, tag)
            assert labels[i] == str(g[f"u{i}_{tag}_labels"]), (i, tag)


def test_g4_device_label_pass():
    """The device label pass (the code that writes the decoders' label records) over the hand-written G4 paths:
    the reference's strings exactly, with and without room for every label in the record."""
    from loe_speech_recognition_amd import HiddenMarkovModelInference, _native

    g4 = load("hmm_g4_labels.npz")
    inf = loop_decoder(load("hmm_g3_loop.npz"))          # uploads the 12-word graph (5 states per word, 3 for S)
    eng = inf._bind()
    names = inf._model_boundaries._labels
    n = int(g4["n"])
    paths = [g4[f"p{i}"].astype(np.int8) for i in range(n)]
    for cap in (32, 2):
        labels, count = eng.labels_from_paths(paths, label_cap=cap)
        labels, count = labels.cpu().numpy(), count.cpu().numpy()
        for i in range(n):
            want = str(g4[f"l{i}"])
            assert count[i] == len(want), (i, want)
            assert "".join(names[w] for w in labels[i, :min(cap, count[i])]) == want[:cap], (i, cap)
    # a path holding -1 (one-frame utterance) or a state outside the graph emits nothing for those frames
    labels, count = eng.labels_from_paths([np.array([-1], np.int8), np.array([0, 0, 100, 5], np.int8)])
    assert count.cpu().numpy().tolist() == [0, 1]          # the change of word after the foreign state has no valid predecessor


def test_dense_and_triangular_whitening_decode_identically(monkeypatch):
    """2 048 synthetic clips decoded with SciPy's whitening matrices uploaded unchanged (all 100 blocks per state)
    and with their lower-triangular rotation (55 blocks, the default): identical scores, paths and strings."""
    from loe_speech_recognition_amd import HiddenMarkovModelInference, _native, synthetic

    models = synthetic.load_word_models()
    sigs = []
    for cfg, count, digits in ((2, 512, 1), (3, 1024, 4), (5, 512, 7)):
        sigs += synthetic.batch(cfg, count, digits, start=5000)[0]
    results = []
    for dense in ("1", "0"):
        monkeypatch.setenv("LSR_DENSE_WHITEN", dense)
        inf = HiddenMarkovModelInference.from_models([models[lab] for lab in synthetic.LOOP_ORDER])
        inf._log_transition_probability_between_words = -100
        eng = inf._bind()
        fb = eng.mfcc(eng.audio_batch(sigs))
        scores, paths, labels, n_labels = eng.decode_loop(fb, want_paths=True)
        results.append((scores.cpu().numpy(), paths.cpu().numpy(), inf._labels_from_device(eng, fb, labels, n_labels, 32)))
    (s0, p0, l0), (s1, p1, l1) = results
    assert np.array_equal(s0.view(np.uint32), s1.view(np.uint32))
    assert np.array_equal(p0, p1) and l0 == l1 and len(l0) == 2048


def test_g5_isolated_train_step():
    g = load("hmm_g5_training.npz")
    m = trainable("w", g["a_means0"], g["a_cov0"], g["a_trans0"])
    m._train([g[f"a_x{i}"] for i in range(int(g["a_n"]))])
    assert rel_err(m._means, g["a_means1"]) <= 1e-4
    assert rel_err(m._covariances, g["a_cov1"]) <= 1e-4
    assert trans_close(m._transition_probs.to_array(), g["a_trans1"])


def test_g5_from_data():
    from loe_speech_recognition_amd import HiddenMarkovModelTrainable

    g = load("hmm_g5_training.npz")
    utts = [g[f"b_x{i}"] for i in range(int(g["b_n"]))]
    m = HiddenMarkovModelTrainable.from_data("w", utts, num_of_states=5, max_iterations=6, isTqdm=False)
    assert rel_err(m._means, g["b_means"]) <= 1e-4
    assert rel_err(m._covariances, g["b_cov"]) <= 1e-4
    assert trans_close(m._transition_probs.to_array(), g["b_trans"])
    score, path = m.predict(utts[0])
    assert path[0] == 0 and np.all(np.diff(path) >= 0)


def _continuous(g):
    from loe_speech_recognition_amd import HiddenMarkovModelTrainContinuous

    tc = HiddenMarkovModelTrainContinuous(isTqdm=False)
    for lab in "12OS":
        tc._trainable_models[lab] = trainable(lab, g[f"c_{lab}_means0"], g[f"c_{lab}_cov0"], g[f"c_{lab}_trans0"])
    tc._models_loaded = list("12OS")
    labeled = {str(k): [g[f"c_x_{str(k)}_{i}"] for i in range(3)] for k in g["c_keys"]}
    return tc, labeled


def _check_continuous(tc, g):
    for lab in "12OS":
        t = tc._trainable_models[lab]
        assert rel_err(t._means, g[f"c_{lab}_means1"]) <= 1e-4, lab
        assert rel_err(t._covariances, g[f"c_{lab}_cov1"]) <= 1e-4, lab
        assert trans_close(t._transition_probs.to_array(), g[f"c_{lab}_trans1"]), lab


def test_g5_continuous_iteration_fused():
    g = load("hmm_g5_training.npz")
    tc, labeled = _continuous(g)
    tc._iteration(labeled)
    _check_continuous(tc, g)


def test_g5_continuous_iteration_reference_shaped():
    g = load("hmm_g5_training.npz")
    tc, labeled = _continuous(g)
    remuxed = tc._train(labeled)
    for lab in "12OS":
        assert len(remuxed[lab]) == int(g[f"c_{lab}_nseg"])
        assert np.array_equal(np.array([s.signal.shape[0] for s in remuxed[lab]]), g[f"c_{lab}_seg_frames"])
        assert np.array_equal(np.concatenate([s.path for s in remuxed[lab]]), g[f"c_{lab}_seg_paths"])
    tc._update_trainable_model_parameters(remuxed)
    _check_continuous(tc, g)


def test_g6_model_collection():
    from loe_speech_recognition_amd import ModelCollection

    g = load("hmm_g6_collection.npz")
    labels = ["1", "2", "3", "4", "5", "6", "7", "8", "9", "O", "Z"]
    mc = ModelCollection()
    mc._models = [word(g, lab) for lab in labels]
    xs = [g[f"x{i}"] for i in range(5)]
    assert mc.predict_batch(xs) == [str(p) for p in g["preds"]]
    scores = mc.scores_batch(xs)
    for i in range(5):
        assert np.array_equal(scores[i], g[f"scores{i}"])
    assert mc.predict(xs[1]) == "2"          # "8" is a copy of "2": the earlier model wins the tie
