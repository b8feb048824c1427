"""The reference's driver-script flows (train on silence-stripped clips -> connected-digit test ->
penalty sweep) run end to end on the device path over a miniature synthetic corpus on disk."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SCRIPTS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "scripts")


def load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(SCRIPTS, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_train_test_sweep(tmp_path):
    corpus = load("make_synthetic_corpus").write_corpus(str(tmp_path / "corpus"))
    models = load("project5_train_no_empty").main(corpus, str(tmp_path / "cache"), max_iterations=30)
    assert sorted(os.listdir(models)) == ["1", "2", "3", "4", "5", "6", "7", "8", "9", "O", "S", "Z"]
    test = load("project5_test_ndigits_with_sil")
    for n in (2, 4):
        truth, prediction, accuracy = test.main(corpus, models, n=n)
        assert len(truth) == 12 and accuracy >= 0.9, (n, list(zip(truth, prediction)))
    penalties, results = load("project5_find_trans_ndigits_with_sil").main(corpus, models, penalties=[0, -50, -100, -1e5])
    assert len(results) == 4 and results[2] >= 90.0 and results[3] == 0.0      # -1e5 forbids leaving the first word

    # isolated classification of the seen and the unseen split with the eleven word models (project3_predict_simple)
    simple = load("project3_predict_simple").main(corpus, models)
    for name in ("seen", "unseen"):
        truth, pred, accuracy = simple[name]
        assert len(truth) == 11 * 12 and set(truth) == set("123456789OZ") and accuracy >= 0.95, (name, accuracy)
    # per-clip calls agree with the batched ones
    from loe_speech_recognition_amd import MFCC, ModelCollection, TIDigits
    mc = ModelCollection.load_from_files(models)
    clips = TIDigits(corpus, isLazyLoading=True).test_dataset["7"][:3]
    assert [mc.predict(MFCC(c, 16000).feature_vector.T) for c in clips] == simple["unseen"][1][6 * 12:6 * 12 + 3]

    # continuous re-training on 2- and 4-digit training clips (project6_train), from the packaged word models:
    # the re-estimated models are saved and still decode the test strings
    from loe_speech_recognition_amd import HiddenMarkovModelTrainable, synthetic
    strings = load("make_synthetic_corpus").write_corpus(str(tmp_path / "strings"), isolated_per_word=1,
                                                         strings_per_length=80, seed=11)
    for model in synthetic.load_word_models().values():
        model.save(str(tmp_path / "cache" / "packaged"))
    project6 = load("project6_train")
    trained = project6.main(strings, str(tmp_path / "cache"), model_name="packaged", max_iterations=3)
    assert trained.endswith("packaged_continuous_2")
    assert sorted(os.listdir(trained)) == ["1", "2", "3", "4", "5", "6", "7", "8", "9", "O", "S", "Z"]
    truth, prediction, accuracy = test.main(corpus, trained, n=4)
    assert accuracy >= 0.9, list(zip(truth, prediction))
    before = np.load(os.path.join(str(tmp_path / "cache" / "packaged"), "4", "model.npz"))["mean"]
    assert not np.array_equal(np.load(os.path.join(trained, "4", "model.npz"))["mean"], before)
    # The weak models trained above on a dozen silence-stripped clips per word leave a state without frames in the
    # forced alignment (the free cross-word transitions skip it): HMMTrainMeanFail propagates as in the reference,
    # and the driver's try / finally still saves the models
    with pytest.raises(HiddenMarkovModelTrainable.HMMTrainMeanFail):
        project6.main(strings, str(tmp_path / "cache"), max_iterations=3)
    assert len(os.listdir(str(tmp_path / "cache" / "big_model_speech_only_continuous_2"))) == 12


def test_flows_without_silence_model(tmp_path):
    """project3_train -> project5_test_1digit / project5_test_ndigits_no_sil / project5_find_trans_ndigits_no_sil: the
    eleven digit models trained on the raw one-digit clips (leading and trailing silence included), decoded in a free
    loop of 55 states without the silence word."""
    corpus = load("make_synthetic_corpus").write_corpus(str(tmp_path / "corpus"))
    models = load("project3_train").main(corpus, str(tmp_path / "cache"), max_iterations=30)
    assert models.endswith("big_model_no_silence")
    assert sorted(os.listdir(models)) == ["1", "2", "3", "4", "5", "6", "7", "8", "9", "O", "Z"]
    truth, prediction, accuracy = load("project5_test_1digit").main(corpus, models)
    assert len(truth) == 11 * 12 and accuracy >= 0.9, accuracy
    no_sil = load("project5_test_ndigits_no_sil")
    for split in ("train", "test"):
        truth, prediction, accuracy = no_sil.main(corpus, models, n=2, split=split)
        assert len(truth) == 12 and all(set(p) <= set("123456789OZ") for p in prediction)
        assert accuracy >= 0.75, (split, list(zip(truth, prediction)))
    penalties, results = load("project5_find_trans_ndigits_no_sil").main(corpus, models, penalties=[0, -250, -1e5])
    assert penalties == [0, -250, -1e5] and len(results) == 3 and results[1] >= 75.0 and results[2] == 0.0
    # the sweep decodes exactly what a decoder with that penalty decodes
    from loe_speech_recognition_amd import MFCC, TI_DIGITS_LABELS, HiddenMarkovModelInference, TIDigits
    decoder = HiddenMarkovModelInference.from_folder(models, list(TI_DIGITS_LABELS))
    clips = TIDigits(corpus, isLazyLoading=True).train_dataset.get_all_n_digits(2)
    mfccs = MFCC.batch([s for signals in clips.values() for s in signals[:5]], sample_rate=16000)
    rows = decoder.predict_sweep(mfccs, [-250])
    decoder._log_transition_probability_between_words = -250
    assert rows[0] == decoder.predict_batch(mfccs)


def test_file_based_project4_and_project6_flows(tmp_path):
    """scripts/project4_2digits.py, project4_phone.py (concatenated one-digit clips -> MFCC -> isolated classifier) and
    the wav half of project6_interactive.py, each compared with the oracle's isolated classifier / loop decoder run
    on the device's features of the same clips."""
    import scipy.io.wavfile

    from loe_speech_recognition_amd import MFCC, TIDigits, synthetic
    from oracle import hmm_oracle as O

    corpus = load("make_synthetic_corpus").write_corpus(str(tmp_path / "corpus"), isolated_per_word=4, strings_per_length=3)
    models = str(tmp_path / "cache" / "packaged")
    for model in synthetic.load_word_models().values():
        model.save(models)
    arrays = synthetic.load_word_arrays()
    iso = [O.WordModelRef(lab, *arrays[lab]) for lab in synthetic.WORDS]
    train = TIDigits(corpus, isLazyLoading=True).train_dataset
    for name, n_digits, n_samples in (("project4_2digits", 2, 10), ("project4_phone", 4, 50)):
        truth, pred = load(name).main(corpus, models)
        assert len(truth) == len(pred) == n_samples and all(len(t) == n_digits for t in truth)
        assert all(p in synthetic.WORDS for p in pred)                  # one label for the whole concatenation
        feats = MFCC.batch([train.get_combined(t, key=2) for t in truth], 16000)
        assert pred == [O.classify_isolated(iso, x) for x in feats], name
    # the same random draw on every run with the same seed, another one with another seed
    assert load("project4_2digits").main(corpus, models)[0] != load("project4_2digits").main(corpus, models, seed=5)[0]

    # project6_interactive on recordings from the corpus: equal to the oracle's loop decode of the device features
    words = [O.WordModelRef(lab, *arrays[lab]) for lab in synthetic.LOOP_ORDER]
    ltp, gauss, bounds = O.stack_models(words)
    mu, U, ld = O.export_gaussians(gauss)
    band = O.band_from_ltp(ltp, bounds.num_states, [bounds.find_lower(s) for s in range(bounds.num_states)])
    interactive = load("project6_interactive")
    wavs = []
    for dirpath, _, files in os.walk(corpus):
        wavs += [os.path.join(dirpath, f) for f in files if len(f.split(".")[0]) == 5 and "TEST" in dirpath]   # 4 digits + take
    assert len(wavs) >= 3
    right = 0
    for path in sorted(wavs)[:3]:
        got = interactive.main(path, models)
        sr, pcm = scipy.io.wavfile.read(path)
        x = MFCC(pcm.astype(np.float32), sr).feature_vector.T
        _, p = O.viterbi_loop_fast(O.emission_fast(x, mu, U, ld), band, bounds.lower, bounds.upper, -100)
        assert got == "".join(bounds.get_labels(p)), path
        right += got == os.path.basename(path)[:4]
    assert right >= 2
