"""Drop-in surface on the device: persistence round trips, pickling, error behaviour of the C ABI,
size limits, and training on synthetic speech against the oracle's trainers."""
import ctypes
import pickle

import numpy as np
import pytest

from helpers import rel_err, trainable, trans_close
from oracle import hmm_oracle as O
from oracle import mfcc_oracle as M

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def words():
    from loe_speech_recognition_amd import synthetic

    return synthetic.load_word_models()


def test_save_and_from_folder_round_trip(tmp_path, words):
    from loe_speech_recognition_amd import (MFCC, HiddenMarkovModel, HiddenMarkovModelInference,
                                            HiddenMarkovModelTrainContinuous, ModelCollection, synthetic)

    for m in words.values():
        m.save(str(tmp_path))
    sigs, truth = synthetic.batch(3, 6, 3, start=7000)
    feats = MFCC.batch(sigs, 16000)
    direct = HiddenMarkovModelInference.from_models([words[lab] for lab in synthetic.LOOP_ORDER])
    loaded = HiddenMarkovModelInference.from_folder(str(tmp_path), synthetic.WORDS + ["S"])
    assert loaded._model_boundaries._labels == synthetic.LOOP_ORDER           # sorted(os.listdir) order
    for inf in (direct, loaded):
        inf._log_transition_probability_between_words = -100
    assert loaded.predict_batch(feats) == direct.predict_batch(feats) == truth
    one = HiddenMarkovModel.from_folder(str(tmp_path / "4"))
    s0, p0 = words["4"].predict(feats[0])
    s1, p1 = one.predict(feats[0])
    assert s0 == s1 and np.array_equal(p0, p1)
    mc = ModelCollection.load_from_files(str(tmp_path))
    assert [str(m) for m in mc._models] == list(synthetic.WORDS)
    iso, t_iso = synthetic.batch(2, 11, 1, start=300)
    # isolated classification of one-digit clips cut to the word (silence belongs to no digit model)
    tc = HiddenMarkovModelTrainContinuous.from_folder(str(tmp_path), synthetic.WORDS + ["S"])
    assert sorted(tc._trainable_models) == sorted(synthetic.WORDS + ["S"])
    assert np.all(tc._trainable_models["S"]._means == 0) and tc._trainable_models["S"].num_of_states == 3
    with pytest.raises(FileNotFoundError):
        HiddenMarkovModel.from_folder(str(tmp_path / "nope"))


def test_objects_pickle_and_keep_working(words):
    from loe_speech_recognition_amd import MFCC, HiddenMarkovModelInference, ModelCollection, synthetic

    inf = HiddenMarkovModelInference.from_models([words[lab] for lab in synthetic.LOOP_ORDER])
    inf._log_transition_probability_between_words = -100
    sigs, truth = synthetic.batch(3, 4, 2, start=8000)
    feats = MFCC.batch(sigs, 16000)
    before = inf.predict_batch(feats)
    clone = pickle.loads(pickle.dumps(inf))
    assert clone.predict_batch(feats) == before == truth
    mc = ModelCollection()
    mc._models = [words[lab] for lab in synthetic.WORDS]
    mc.predict(feats[0])
    assert pickle.loads(pickle.dumps(mc)).scores_batch(feats).shape == (4, 11)
    # a continuous trainer that has iterated (device batch, chain tables, factorisation caches) pickles to its models
    from loe_speech_recognition_amd import HiddenMarkovModelTrainContinuous

    tc = HiddenMarkovModelTrainContinuous(isTqdm=False)
    for lab in synthetic.WORDS + ["S"]:
        tc.add_model(pickle.loads(pickle.dumps(words[lab])))
    more, more_truth = synthetic.batch(3, 64, 4, start=8100)               # every word occurs: no state stays without frames
    assert set("".join(more_truth)) == set(synthetic.WORDS)
    labeled = synthetic.group_by_label(MFCC.batch(more, 16000), more_truth)
    tc._iteration(labeled)
    clone = pickle.loads(pickle.dumps(tc))
    assert "_chain_cache" not in clone.__dict__ and "_data_cache" not in clone.__dict__
    tc._iteration(labeled)
    clone._iteration(labeled)
    for lab in tc._models_loaded:
        assert np.array_equal(tc._trainable_models[lab]._means, clone._trainable_models[lab]._means), lab


def test_isolated_classifier_on_word_clips(words):
    from loe_speech_recognition_amd import MFCC, ModelCollection, synthetic

    mc = ModelCollection()
    mc._models = [words[lab] for lab in synthetic.WORDS]
    rng = np.random.default_rng(21)
    labels = [synthetic.WORDS[i % 11] for i in range(44)]
    feats = MFCC.batch([synthetic.word_clip(lab, rng) for lab in labels], 16000)
    got = mc.predict_batch(feats)
    arrays = synthetic.load_word_arrays()
    ref_models = [O.WordModelRef(lab, *arrays[lab]) for lab in synthetic.WORDS]
    assert got == [O.classify_isolated(ref_models, x) for x in feats]
    assert np.mean([a == b for a, b in zip(got, labels)]) > 0.9


def test_kernel_timing_can_be_limited_to_named_kernels(words):
    """lsr_profile_enable(LSR_PROFILE_ONLY(k)): only the launches of kernel k carry event pairs."""
    from loe_speech_recognition_amd import HiddenMarkovModelInference, _native, synthetic

    eng = _native.get_engine()
    inf = HiddenMarkovModelInference.from_models([words[lab] for lab in synthetic.LOOP_ORDER])
    inf._log_transition_probability_between_words = -100
    inf._bind()
    sigs, truth = synthetic.batch(3, 8, 2, start=8300)
    ab = eng.audio_batch(sigs)
    try:
        eng.profile_reset()
        eng.profile_enable(True, only=("emission", "viterbi"))
        for _ in range(3):
            eng.recognize(ab, want_paths=False, label_cap=16)
        counts = {k: eng.profile_get(k) for k in ("frame_index", "mel_power", "mfcc_finish", "emission", "viterbi")}
        assert counts["emission"][1] == 3 and counts["viterbi"][1] == 3
        assert counts["emission"][0] > 0 and counts["viterbi"][0] > 0
        assert counts["frame_index"][1] == counts["mel_power"][1] == counts["mfcc_finish"][1] == 0
        eng.profile_reset()
        eng.profile_enable(True)
        eng.recognize(ab, want_paths=False, label_cap=16)
        assert all(eng.profile_get(k)[1] == 1 for k in counts)
    finally:
        eng.profile_enable(False)
        eng.profile_reset()


def test_host_layer_checks_shapes_before_the_kernels_see_them(words):
    """Shape mismatches are ValueErrors raised on the host (not asserts: the kernels index rows of 39 floats and trust
    the sizes they are given)."""
    from loe_speech_recognition_amd import _native

    eng = _native.get_engine()
    with pytest.raises(ValueError, match="shape"):
        eng.feature_batch([np.zeros((10, 38), np.float32)])
    with pytest.raises(ValueError, match="shape"):
        eng.feature_batch([np.zeros((39,), np.float32)])
    with pytest.raises(ValueError, match="model arrays"):
        eng.set_model(_native.new_token(), np.zeros((3, 39)), np.zeros((3, 39, 38)), np.zeros(3))
    with pytest.raises(ValueError, match="model arrays"):
        eng.set_model(_native.new_token(), np.zeros((3, 39)), np.zeros((2, 39, 39)), np.zeros(3))


def test_c_abi_rejects_bad_arguments():
    from loe_speech_recognition_amd import _native

    eng = _native.get_engine()
    lib = eng.lib
    null = ctypes.c_void_p(0)
    assert lib.lsr_mfcc_batch(eng.ctx, null, null, null, 1, 10, null) == -1
    assert b"null" in lib.lsr_last_error(eng.ctx)
    d = (ctypes.c_double * 4)()
    assert lib.lsr_model_upload(eng.ctx, 0, d, d, d) == -1
    assert lib.lsr_model_upload(eng.ctx, 5000, d, d, d) == -3
    band = np.zeros((3, 130), np.float32)
    lo = np.array([0], np.int32)
    up = np.array([129], np.int32)
    rc = lib.lsr_loop_graph_upload(eng.ctx, 130, 1, band.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                   lo.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                   up.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), -1)
    assert rc == -3 and b"127" in lib.lsr_last_error(eng.ctx)
    assert lib.lsr_loop_set_between(eng.ctx, ctypes.c_double(float("nan")), 0) == -1
    assert lib.lsr_profile_get(eng.ctx, 99, None, None) == -1
    fresh = ctypes.c_void_p()
    assert lib.lsr_ctx_create(0, ctypes.byref(fresh)) == 0
    x = eng.empty((64, 39), eng.torch.float32)
    assert lib.lsr_emission_batch(fresh, ctypes.c_void_p(x.data_ptr()), 64, ctypes.c_void_p(x.data_ptr())) == -4   # no model yet
    assert lib.lsr_ctx_destroy(fresh) == 0
    assert lib.lsr_ctx_create(99, ctypes.byref(fresh)) != 0
    # entry points of the rows around the path
    p = ctypes.c_void_p(x.data_ptr())
    assert lib.lsr_separation_scan(eng.ctx, null, 0, p, p, 1, 10, 160, 0.06, 0.01, 2, 9, p, p, p, p, null, null) == -1
    assert lib.lsr_separation_scan(eng.ctx, p, 0, p, p, 1, 10, 300, 0.06, 0.01, 2, 9, p, p, p, p, null, null) == -3
    assert b"frame_size" in lib.lsr_last_error(eng.ctx)
    assert lib.lsr_separation_scan(eng.ctx, p, 0, p, p, 0, 10, 160, 0.06, 0.01, 2, 9, p, p, p, p, null, null) == -1
    assert lib.lsr_gather_segments(eng.ctx, p, 3, p, p, 1, 10, p) == -1          # element size must be 2 or 4
    assert lib.lsr_gather_segments(eng.ctx, p, 4, p, p, 0, 10, p) == -1
    pen = (ctypes.c_double * 2)(-100.0, float("nan"))
    wide = (ctypes.c_int32 * 2)(0, 0)
    assert lib.lsr_decode_loop_sweep(eng.ctx, p, p, 1, 64, 64, pen, wide, 2, null, p, p, 16) == -1
    assert b"NaN" in lib.lsr_last_error(eng.ctx)
    assert lib.lsr_decode_loop_sweep(eng.ctx, p, p, 1, 64, 64, pen, wide, 0, null, p, p, 16) == -1
    assert lib.lsr_accumulate_stats(eng.ctx, p, p, 64, 0, p, p, p) == -3
    assert lib.lsr_accumulate_stats(eng.ctx, p, p, 64, 5000, p, p, p) == -3


def test_sixty_four_state_chain_and_limit():
    """Largest one-wavefront alignment chain (64 positions) and the first two-wavefront sizes (72, 127) against the
    oracle; 128 and more are refused (the reference's int8 pointers end at 127)."""
    from loe_speech_recognition_amd import HiddenMarkovModelMultiWord

    rng = np.random.default_rng(33)
    parts = {}
    for lab, n in (("a", 5), ("b", 5), ("c", 4), ("S", 3)):
        means = (rng.standard_normal((n, 39)) * 2).astype(np.float32)
        cov = np.stack([(lambda a: (a @ a.T / 39 + 1e-2 * np.eye(39)))(rng.standard_normal((39, 39))) for _ in range(n)]).astype(np.float32)
        trans = np.zeros((n, n), np.float32)
        for s in range(n):
            w = rng.uniform(0.2, 1, size=min(3, n - s))
            trans[s, s:s + len(w)] = w / w.sum()
        parts[lab] = (means, cov, trans)
    prod = {lab: trainable(lab, *p) for lab, p in parts.items()}
    chain = "SaSbSaSbSaScScScS"                       # 9*3 + 5*5 + 3*4 = 64 states
    n_states = sum(parts[c][0].shape[0] for c in chain)
    assert n_states == 64
    hmm = HiddenMarkovModelMultiWord.from_labels(chain, prod)
    seq = np.concatenate([parts[c][0] for c in chain])
    idx = np.minimum(np.arange(200) * 64 // 200, 63)
    x = (seq[idx] + 0.5 * rng.standard_normal((200, 39))).astype(np.float32)
    score, path = hmm.predict(x)
    ltp, gauss, _ = O.stack_models([O.WordModelRef(c, *parts[c]) for c in chain])
    mu, U, ld = O.export_gaussians(gauss)
    s_ref, p_ref, _ = O.viterbi_chain_fast(O.emission_fast(x, mu, U, ld), O.band_from_ltp(ltp, 64))
    assert score == s_ref and np.array_equal(path, p_ref)
    for longer in (chain + "aS", chain + "aSbSaSbSaSbSaScS"):          # 72 and 127 states
        hmm = HiddenMarkovModelMultiWord.from_labels(longer, prod)
        n = hmm.num_of_states
        seq = np.concatenate([parts[c][0] for c in longer])
        idx = np.minimum(np.arange(300) * n // 300, n - 1)
        xl = (seq[idx] + 0.5 * rng.standard_normal((300, 39))).astype(np.float32)
        ltp, gauss, _ = O.stack_models([O.WordModelRef(c, *parts[c]) for c in longer])
        mu, U, ld = O.export_gaussians(gauss)
        # a mixed batch: the long utterance, a short one, and the 64-state input again (all take the wide form)
        got = hmm.predict_batch([xl, xl[:n + 3], xl[:150]])
        for (score, path), xi in zip(got, (xl, xl[:n + 3], xl[:150])):
            s_ref, p_ref, _ = O.viterbi_chain_fast(O.emission_fast(xi, mu, U, ld), O.band_from_ltp(ltp, n))
            assert score == s_ref and np.array_equal(path, p_ref), n
    assert n == 127
    too_long = HiddenMarkovModelMultiWord.from_labels(chain + "aSbSaSbSaSbSaScSaS", prod)
    with pytest.raises(ValueError):
        too_long.predict(x)


def test_isolated_training_on_synthetic_words_matches_oracle():
    from loe_speech_recognition_amd import MFCC, HiddenMarkovModelTrainable, synthetic

    rng = np.random.default_rng(5)
    utts = MFCC.batch([synthetic.word_clip("7", rng) for _ in range(20)], 16000)
    try:
        ref = O.WordModelRef.from_data("7", utts, n_states=5, max_iterations=8)
    except O.EmptyStateError:
        with pytest.raises(HiddenMarkovModelTrainable.HMMTrainMeanFail):
            HiddenMarkovModelTrainable.from_data("7", utts, num_of_states=5, max_iterations=8, isTqdm=False)
        return
    got = HiddenMarkovModelTrainable.from_data("7", utts, num_of_states=5, max_iterations=8, isTqdm=False)
    assert rel_err(got._means, ref.means) <= 1e-4
    assert rel_err(got._covariances, ref.cov) <= 1e-4
    assert trans_close(got._transition_probs.to_array(), ref.trans)


def test_continuous_iteration_on_synthetic_speech_matches_oracle(words):
    """Config 4 at reduced size: 72 clips of 1-7 digits, one forced-alignment + re-estimation pass."""
    from loe_speech_recognition_amd import MFCC, HiddenMarkovModelTrainContinuous, synthetic

    sigs, truth = synthetic.batch(4, 72, [1, 2, 3, 4, 5, 7])
    feats = MFCC.batch(sigs, 16000)
    labeled = {}
    for x, d in zip(feats, truth):
        labeled.setdefault(d, []).append(x)
    arrays = synthetic.load_word_arrays()
    order = synthetic.WORDS + ["S"]
    ref = {lab: O.WordModelRef(lab, *arrays[lab]) for lab in order}
    for m in ref.values():                       # the trainer restarts means at zero (hidden_markov_model.py:696-705)
        m.means = np.zeros_like(m.means)
    try:
        O.continuous_iteration(ref, labeled)
        failed = False
    except O.EmptyStateError:
        failed = True
    tc = HiddenMarkovModelTrainContinuous(isTqdm=False)
    fresh = synthetic.load_word_models()
    for lab in order:
        tc.add_model(fresh[lab])
    if failed:
        with pytest.raises(Exception):
            tc._iteration(labeled)
        return
    tc._iteration(labeled)
    for lab in order:
        t = tc._trainable_models[lab]
        assert rel_err(t._means, ref[lab].means) <= 1e-4, lab
        assert rel_err(t._covariances, ref[lab].cov) <= 1e-4, lab
        assert trans_close(t._transition_probs.to_array(), ref[lab].trans), lab
    # train() drives the same iteration and stops on convergence or after max_iterations
    tc.train(labeled, max_iterations=2)


@pytest.mark.parametrize("n_mix", [2, 4])
def test_gaussian_mixture_emissions_and_decode(n_mix):
    """Mixture emissions (log-sum over components) against the oracle's restatement of
    deprecated/gaussian_mixture_model.py:159-164 (itself pinned by the reference's values in gmm_g9_emission.npz),
    and a loop decode that uses them."""
    from loe_speech_recognition_amd import (GaussianMixture, HiddenMarkovModel, HiddenMarkovModelInference,
                                            LogTransitionProbabilities, TransitionProbabilities, _native)
    from loe_speech_recognition_amd.hidden_markov_model import stack_gaussians

    rng = np.random.default_rng(40 + n_mix)
    labels = ["1", "2", "S"]
    sizes = [5, 5, 3]
    weights, o_comps, mixtures = [], [], []
    for _ in range(sum(sizes)):
        w = rng.dirichlet(np.ones(n_mix) * 2)
        if len(weights) == 3:
            w = np.concatenate([[0.0], w[1:] / w[1:].sum()])           # a zero weight: log w = -inf
        means = (rng.standard_normal((n_mix, 39)) * 2).astype(np.float32)
        cov = np.stack([(lambda a: a @ a.T / 39 + 1e-2 * np.eye(39))(rng.standard_normal((39, 39))) for _ in range(n_mix)]).astype(np.float32)
        weights.append(w)
        o_comps.append([O.GaussianRef(m, c) for m, c in zip(means, cov)])
        mixtures.append(GaussianMixture.from_weights_means_covariances(w, means, cov))
    x = (rng.standard_normal((150, 39)) * 2).astype(np.float32)
    eng = _native.get_engine()
    eng.set_model(None, *stack_gaussians(mixtures))
    got = eng.emission(eng.feature_batch([x])).cpu().numpy()[:150, :13]
    ref = O.emission_gmm_fast(x, np.array(weights), o_comps)
    assert rel_err(got, ref) <= 1e-4
    assert np.all(np.abs(got - ref) <= 2 * np.spacing(np.abs(ref)))
    assert np.mean(got != ref) < 0.02
    slow = O.emission_gmm_ref(x[:3], np.array(weights), o_comps)
    assert np.all(np.abs(got[:3] - slow) <= 2 * np.spacing(np.abs(slow)))
    v = mixtures[5].log_pdf(x[7])
    assert v.dtype == np.float32 and abs(v - ref[7, 5]) <= 2 * np.spacing(abs(ref[7, 5]))
    # decode with mixture states: the recursion sees the same float32 scores as the oracle recursion
    models, at = [], 0
    trans = {}
    for lab, n in zip(labels, sizes):
        t = np.zeros((n, n), np.float32)
        for s in range(n):
            w = rng.uniform(0.2, 1, size=min(3, n - s))
            t[s, s:s + len(w)] = w / w.sum()
        trans[lab] = t
        m = HiddenMarkovModel(lab)
        m._multivariate_normals = mixtures[at:at + n]
        m._log_transition_probs = LogTransitionProbabilities.from_transition_probability(
            TransitionProbabilities.from_transition_probability(t))
        models.append(m)
        at += n
    inf = HiddenMarkovModelInference.from_models(models)
    inf._log_transition_probability_between_words = -20
    score, path = inf._viterbi(x)
    ltp = O.LogTransRef()
    for lab in labels:
        ltp.append(O.LogTransRef.from_probs(trans[lab]))
    bounds = O.BoundariesRef(sizes, labels)
    band = O.band_from_ltp(ltp, 13, [bounds.find_lower(s) for s in range(13)])
    s_ref, p_ref = O.viterbi_loop_fast(got, band, bounds.lower, bounds.upper, -20)
    assert score == s_ref and np.array_equal(path, p_ref)
    assert inf.predict(x) == "".join(bounds.get_labels(p_ref))
    # chain decode and persistence keep the mixtures
    one = models[0].predict(x[:40])
    s1, p1, _ = O.viterbi_chain_fast(got[:40, :5], O.band_from_ltp(O.LogTransRef.from_probs(trans["1"]), 5))
    assert one[0] == s1 and np.array_equal(one[1], p1)


def test_resident_model_slots(words):
    """lsr_model_slot: several uploaded models stay on the device and a switch selects one without a new upload.
    Emission scores after switching back and forth equal those of a fresh upload of the same model, an empty slot
    refuses to compute, and the engine's token -> slot table survives more models than slots (least recently used
    goes first)."""
    import ctypes

    from loe_speech_recognition_amd import _native
    from loe_speech_recognition_amd.hidden_markov_model import stack_gaussians

    eng = _native.get_engine()
    rng = np.random.default_rng(17)
    x = [(1.5 * rng.standard_normal((40, 39))).astype(np.float32)]
    fb = eng.feature_batch(x)
    labels = sorted(words)
    sets = [stack_gaussians(words[lab]._multivariate_normals) for lab in labels]
    fresh = []
    for arrays in sets:                                     # reference results: anonymous uploads (slot 0)
        eng.set_model(None, *arrays)
        fresh.append(eng.emission(fb).cpu().numpy())
    tokens = [_native.new_token() for _ in sets]
    for rounds in range(2):                                 # more models than slots: eviction and re-upload on the way
        for tok, arrays, ref in zip(tokens, sets, fresh):
            eng.set_model(tok, *arrays)
            assert np.array_equal(eng.emission(fb).cpu().numpy(), ref, equal_nan=True)
    assert len(eng._resident) <= eng.MODEL_SLOTS - 1
    # the last models are resident: selecting them uploads nothing (the arrays are not even looked at)
    for tok, ref in list(zip(tokens, fresh))[-3:]:
        eng.set_model(tok, None, None, None)
        assert np.array_equal(eng.emission(fb).cpu().numpy(), ref, equal_nan=True)
    # C ABI: slot range is checked, an empty slot holds no model
    assert eng.lib.lsr_model_slot(eng.ctx, 8) == -1 and eng.lib.lsr_model_slot(eng.ctx, -1) == -1
    used = {v[0] for v in eng._resident.values()} | {0}
    free = [k for k in range(eng.MODEL_SLOTS) if k not in used]
    if free:
        eng._select_slot(free[0])
        try:
            out = eng.empty((64, 64), eng.torch.float32)
            rc = eng.lib.lsr_emission_batch(eng.ctx, ctypes.c_void_p(fb.feats.data_ptr()), fb.total_frames,
                                            ctypes.c_void_p(out.data_ptr()))
            assert rc == -4                                  # LSR_ERR_STATE: no model uploaded
        finally:
            eng._select_slot(0)
            eng._model_token = None
