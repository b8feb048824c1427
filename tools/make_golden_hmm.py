"""Generate tests/golden/hmm_*.npz by running the REAL reference HMM code.

Runs only in the build container, where /root/reference is mounted.  The
reference package cannot be imported through its __init__ (it needs librosa,
sounddevice and Python >= 3.12 syntax), so the four HMM modules are loaded
directly with three shims: typing.Self, an empty `uniplot` module and an empty
package object whose __path__ points at the reference sources.  Nothing from
the reference is copied: the fixtures hold seeded inputs and the outputs the
reference computed for them.

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden_hmm.py
"""
import importlib
import os
import sys
import types
import typing

import numpy as np
import typing_extensions

sys.dont_write_bytecode = True
typing.Self = typing_extensions.Self
sys.modules["uniplot"] = types.ModuleType("uniplot")
_pkg = types.ModuleType("loe_speech_recognition")
_pkg.__path__ = ["/root/reference/src/loe_speech_recognition"]
sys.modules["loe_speech_recognition"] = _pkg
H = importlib.import_module("loe_speech_recognition.hidden_markov_model")
TP = importlib.import_module("loe_speech_recognition.transition_probability")
MB = importlib.import_module("loe_speech_recognition.model_boundary")
SG = importlib.import_module("loe_speech_recognition.signal")
_ti = importlib.import_module("loe_speech_recognition.ti_digits")
MC = importlib.import_module("loe_speech_recognition.model_collection")

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
D = 39
LOOP_LABELS = ["1", "2", "3", "4", "5", "6", "7", "8", "9", "O", "S", "Z"]   # sorted(os.listdir) order


def random_word(rng, n_states, spread=1.0):
    means = (rng.standard_normal((n_states, D)) * 2.0).astype(np.float32)
    cov = np.empty((n_states, D, D), dtype=np.float32)
    for s in range(n_states):
        a = rng.standard_normal((D, D)) * spread
        cov[s] = (a @ a.T / D + 1e-3 * np.eye(D)).astype(np.float32)
    trans = np.zeros((n_states, n_states), dtype=np.float32)
    for s in range(n_states):
        w = rng.uniform(0.2, 1.0, size=min(3, n_states - s))
        trans[s, s:s + len(w)] = (w / w.sum()).astype(np.float32)
    return means, cov, trans


def make_trainable(label, means, cov, trans):
    m = H.HiddenMarkovModelTrainable(label, isMultiProcessing=False, isTqdm=False)
    m._means = means.copy()
    m._covariances = cov.copy()
    m._transition_probs = TP.TransitionProbabilities.from_transition_probability(trans.copy())
    m._update_inference_weights()
    return m


def sample_utt(rng, means, T, noise=0.7):
    """Frames that walk left to right through the rows of `means`."""
    n = means.shape[0]
    cuts = np.sort(rng.choice(np.arange(1, T), size=n - 1, replace=False)) if T > n else np.arange(1, n)
    state = np.searchsorted(cuts, np.arange(T), side="right")
    state = np.minimum(state, n - 1)
    return (means[state] + noise * rng.standard_normal((T, D))).astype(np.float32)


def dense_trans(tp, n):
    out = np.full((n, n), np.nan, dtype=np.float32)
    for (i, j), v in tp._core.items():
        out[i, j] = v
    return out


def g1_emission(rng):
    means, cov, _ = random_word(rng, 58)
    mvns = H.HiddenMarkovModelTrainable.get_multivariate_normals(means, cov)
    x = (rng.standard_normal((64, D)) * 2.5).astype(np.float32)
    ll = np.empty((64, 58), dtype=np.float32)
    for t in range(64):
        for s in range(58):
            ll[t, s] = mvns[s].log_pdf(x[t])
    U = np.stack([mvns[s]._core.cov_object._LP for s in range(4)])
    logdet = np.array([mvns[s]._core.cov_object._log_pdet for s in range(58)])
    np.savez_compressed(os.path.join(OUT, "hmm_g1_emission.npz"), means=means, cov=cov, x=x, ll=ll,
                        U_first4=U, logdet=logdet)


def g2_single(rng):
    out = {}
    # plain single models
    for n_states in (3, 5):
        means, cov, trans = random_word(rng, n_states)
        m = make_trainable("w", means, cov, trans)
        out[f"n{n_states}_means"], out[f"n{n_states}_cov"], out[f"n{n_states}_trans"] = means, cov, trans
        for T in (9, 50, 200):
            x = sample_utt(rng, means, T)
            score, path = m.predict(x)
            out[f"n{n_states}_T{T}_x"], out[f"n{n_states}_T{T}_score"], out[f"n{n_states}_T{T}_path"] = \
                x, np.float32(score), path
        # last state reachable only on the final frame: exercises the back-trace quirk
        x = sample_utt(rng, means[:n_states - 1], 12)
        x = np.concatenate([x, (means[n_states - 1:] + 0.01).astype(np.float32)])
        score, path = m.predict(x)
        out[f"n{n_states}_late_x"], out[f"n{n_states}_late_score"], out[f"n{n_states}_late_path"] = \
            x, np.float32(score), path
    # alignment chains: words stacked as S d S d ... S (cross-word keys missing -> 0.0)
    words = {}
    for lab in "123S":
        words[lab] = random_word(rng, 3 if lab == "S" else 5)
        out[f"chain_{lab}_means"], out[f"chain_{lab}_cov"], out[f"chain_{lab}_trans"] = words[lab]
    trainables = {lab: make_trainable(lab, *words[lab]) for lab in words}
    for chain, tag in (("S1S2S3S2S", "n35"), ("S1S2S3S1S2S3S1S", "n59")):
        hmm = H.HiddenMarkovModelMultiWord.from_labels(chain, trainables)
        all_means = np.concatenate([words[c][0] for c in chain])
        assert all_means.shape[0] == int(tag[1:])
        out[f"{tag}_chain"] = np.array(chain)
        for T in (50, 200):
            if T <= all_means.shape[0]:
                continue
            x = sample_utt(rng, all_means, T)
            score, path = hmm.predict(x)
            out[f"{tag}_T{T}_x"], out[f"{tag}_T{T}_score"], out[f"{tag}_T{T}_path"] = x, np.float32(score), path
    np.savez_compressed(os.path.join(OUT, "hmm_g2_single.npz"), **out)


def build_loop(words):
    inf = H.HiddenMarkovModelInference()
    ltp = H.LogTransitionProbabilities()
    mvn = []
    mb = MB.ModelBoundary()
    for lab in LOOP_LABELS:
        m = make_trainable(lab, *words[lab])
        ltp.append(m._log_transition_probs)
        mvn.extend(m._multivariate_normals)
        mb.append(m.num_of_states)
    mb.add_model_labels(list(LOOP_LABELS))
    inf._log_transition_probs = ltp
    inf._multivariate_normals = mvn
    inf._model_boundaries = mb
    return inf


def g3_loop(rng):
    out = {}
    words = {lab: random_word(rng, 3 if lab == "S" else 5) for lab in LOOP_LABELS}
    words["7"] = tuple(a.copy() for a in words["3"])        # identical words: exact score ties
    for lab in LOOP_LABELS:
        out[f"{lab}_means"], out[f"{lab}_cov"], out[f"{lab}_trans"] = words[lab]
    inf = build_loop(words)
    assert inf._model_boundaries.lower_boundaries == [0, 5, 10, 15, 20, 25, 30, 35, 40, 45, 50, 53]
    utts = []
    for digits in ("4", "3", "77", "S1S", "90Z5", "S2S2S", "1234567"):
        seq = np.concatenate([words["O" if c == "0" else c][0] for c in digits])
        utts.append(sample_utt(rng, seq, max(12, 14 * len(digits) + int(rng.integers(0, 9)))))
    utts.append((rng.standard_normal((30, D)) * 3).astype(np.float32))         # off-model noise
    for i, x in enumerate(utts):
        out[f"u{i}_x"] = x
        for tag, between in (("m100", -100), ("m250", -250), ("dflt", np.log(0.005)), ("m3f", -3.0)):
            inf._log_transition_probability_between_words = between
            score, path = inf._viterbi(observation_sequence=x)
            out[f"u{i}_{tag}_score"], out[f"u{i}_{tag}_path"] = np.float32(score), path
            out[f"u{i}_{tag}_labels"] = np.array(inf.predict(x))
    out["n_utts"] = np.int64(len(utts))
    np.savez_compressed(os.path.join(OUT, "hmm_g3_loop.npz"), **out)


def g4_labels():
    mb = MB.ModelBoundary()
    for lab in LOOP_LABELS:
        mb.append(3 if lab == "S" else 5)
    mb.add_model_labels(list(LOOP_LABELS))
    paths = [
        [0, 0, 1, 2, 3, 4, 4],
        [50, 50, 51, 52, 15, 16, 17, 18, 19, 19, 50, 51, 52],
        [15, 16, 19, 15, 16, 19, 19],                      # repeated word: last state -> first state
        [15, 16, 18, 15, 16, 19],                          # back to first state NOT from last state: no repeat
        [53, 54, 57, 53, 57, 53],                          # Z Z Z
        [50, 51, 52, 50, 51, 52],                          # silence repeated -> nothing
        [7],
        [52, 52, 52],
        [45, 47, 49, 0, 2, 4, 5, 9, 50, 52, 10, 14, 14, 53, 55, 57],
        [20, 21, 22, 25, 26, 20, 24, 20],
    ]
    out = {"n": np.int64(len(paths))}
    for i, p in enumerate(paths):
        arr = np.array(p, dtype=np.int8)
        out[f"p{i}"] = arr
        out[f"l{i}"] = np.array("".join(mb.get_labels(arr)))
        out[f"k{i}"] = np.array("".join(mb.get_labels(arr, skip_silence=False)))
    np.savez_compressed(os.path.join(OUT, "hmm_g4_labels.npz"), **out)


def g5_training(rng):
    out = {}
    # (a) one isolated-word _train step from hand-set parameters
    means, cov, trans = random_word(rng, 5)
    utts = [sample_utt(rng, means, int(rng.integers(18, 40))) for _ in range(12)]
    m = make_trainable("w", means, cov, trans)
    m._train(utts)
    out["a_means0"], out["a_cov0"], out["a_trans0"] = means, cov, trans
    out["a_n"] = np.int64(len(utts))
    for i, u in enumerate(utts):
        out[f"a_x{i}"] = u
    out["a_means1"], out["a_cov1"] = m._means, m._covariances
    out["a_trans1"] = dense_trans(m._transition_probs, 5)
    # (b) from_data end to end (init from first utterance, iterate to convergence / 6 iterations)
    utts_b = []
    for _ in range(10):
        durs = rng.integers(5, 9, size=5)
        state = np.repeat(np.arange(5), durs)
        utts_b.append((means[state] + 0.3 * rng.standard_normal((state.shape[0], D))).astype(np.float32))
    out["b_n"] = np.int64(len(utts_b))
    for i, u in enumerate(utts_b):
        out[f"b_x{i}"] = u
    m2 = H.HiddenMarkovModelTrainable.from_data("w", utts_b, num_of_states=5, max_iterations=6,
                                                isMultiProcessingTraining=False, isTqdm=False)
    out["b_means"], out["b_cov"] = m2._means, m2._covariances
    out["b_trans"] = dense_trans(m2._transition_probs, 5)
    # (c) one continuous-training iteration over three word models + silence
    words = {lab: random_word(rng, 3 if lab == "S" else 5) for lab in "12OS"}
    tc = H.HiddenMarkovModelTrainContinuous(isTqdm=False, isMultiProcessing=False)
    for lab in "12OS":
        tc._trainable_models[lab] = make_trainable(lab, *words[lab])
        out[f"c_{lab}_means0"], out[f"c_{lab}_cov0"], out[f"c_{lab}_trans0"] = words[lab]
    tc._models_loaded = list("12OS")
    labeled = {}
    for digits in ("12", "2O1", "O", "1O2O"):
        chain = tc.insert_silence(digits)
        seq = np.concatenate([words[c][0] for c in chain])
        labeled[digits] = [sample_utt(rng, seq, int(rng.integers(9, 14)) * len(chain)) for _ in range(3)]
        for i, u in enumerate(labeled[digits]):
            out[f"c_x_{digits}_{i}"] = u
    out["c_keys"] = np.array(list(labeled.keys()))
    remuxed = tc._train(labeled)
    for lab, sigs in remuxed.items():
        out[f"c_{lab}_nseg"] = np.int64(len(sigs))
        out[f"c_{lab}_seg_frames"] = np.array([s.signal.shape[0] for s in sigs], dtype=np.int64)
        out[f"c_{lab}_seg_paths"] = np.concatenate([s.path for s in sigs]).astype(np.int8)
    tc._update_trainable_model_parameters(remuxed)
    for lab in "12OS":
        t = tc._trainable_models[lab]
        out[f"c_{lab}_means1"], out[f"c_{lab}_cov1"] = t._means, t._covariances
        out[f"c_{lab}_trans1"] = dense_trans(t._transition_probs, t.num_of_states)
    np.savez_compressed(os.path.join(OUT, "hmm_g5_training.npz"), **out)


def g6_collection(rng):
    out = {}
    labels = list(_ti.TI_DIGITS_LABELS.keys())
    words = {lab: random_word(rng, 5) for lab in labels}
    words["8"] = tuple(a.copy() for a in words["2"])        # tie: "2" precedes "8" in the label table
    mc = MC.ModelCollection()
    for lab in labels:
        mc._models.append(make_trainable(lab, *words[lab]))
        out[f"{lab}_means"], out[f"{lab}_cov"], out[f"{lab}_trans"] = words[lab]
    preds = []
    for i, lab in enumerate(["2", "8", "5", "Z", "O"]):
        x = sample_utt(rng, words[lab][0], 25 + i)
        out[f"x{i}"] = x
        preds.append(mc.predict(x))
        out[f"scores{i}"] = np.array([m.predict(x)[0] for m in mc._models], dtype=np.float32)
    out["preds"] = np.array(preds)
    np.savez_compressed(os.path.join(OUT, "hmm_g6_collection.npz"), **out)


def g7_wide(rng):
    """Graphs beyond 64 states (the int8 back-pointers of hidden_markov_model.py:498-501 reach 127): forced-alignment
    chains of 75 and 123 states and a free word loop over 24 five-state words plus silence (123 states)."""
    out = {}
    words = {}
    for lab in "123S":
        words[lab] = random_word(rng, 3 if lab == "S" else 5)
        out[f"chain_{lab}_means"], out[f"chain_{lab}_cov"], out[f"chain_{lab}_trans"] = words[lab]
    trainables = {lab: make_trainable(lab, *words[lab]) for lab in words}
    for digits, tag in (("123123123", "n75"), ("312213321123231", "n123")):
        chain = "S" + "".join(c + "S" for c in digits)
        hmm = H.HiddenMarkovModelMultiWord.from_labels(chain, trainables)
        all_means = np.concatenate([words[c][0] for c in chain])
        assert all_means.shape[0] == int(tag[1:])
        out[f"{tag}_chain"] = np.array(chain)
        for T in (160, 300):
            x = sample_utt(rng, all_means, T)
            score, path = hmm.predict(x)
            assert path.dtype == np.int8 and path.max() > 64
            out[f"{tag}_T{T}_x"], out[f"{tag}_T{T}_score"], out[f"{tag}_T{T}_path"] = x, np.float32(score), path
    labels = [chr(ord("a") + i) for i in range(24)]
    order = sorted(labels + ["S"])                                  # "S" sorts before the lower-case names
    loop_words = {lab: random_word(rng, 3 if lab == "S" else 5) for lab in order}
    loop_words["q"] = tuple(a.copy() for a in loop_words["c"])      # identical words on both sides of state 64: ties
    inf = H.HiddenMarkovModelInference()
    ltp = H.LogTransitionProbabilities()
    mvn = []
    mb = MB.ModelBoundary()
    for lab in order:
        m = make_trainable(lab, *loop_words[lab])
        out[f"loop_{lab}_means"], out[f"loop_{lab}_cov"], out[f"loop_{lab}_trans"] = loop_words[lab]
        ltp.append(m._log_transition_probs)
        mvn.extend(m._multivariate_normals)
        mb.append(m.num_of_states)
    mb.add_model_labels(list(order))
    inf._log_transition_probs, inf._multivariate_normals, inf._model_boundaries = ltp, mvn, mb
    assert mb.upper_boundaries[-1] == 122
    out["loop_order"] = np.array(order)
    utts = []
    for text in ("a", "x", "qc", "SbS", "mnop", "wSxSw", "abcxwvq", "cq"):
        seq = np.concatenate([loop_words[c][0] for c in text])
        utts.append(sample_utt(rng, seq, max(12, 13 * len(text) + int(rng.integers(0, 9)))))
    utts.append((rng.standard_normal((25, D)) * 3).astype(np.float32))
    for i, x in enumerate(utts):
        out[f"u{i}_x"] = x
        for tag, between in (("m100", -100), ("dflt", np.log(0.005)), ("m3f", -3.0)):
            inf._log_transition_probability_between_words = between
            score, path = inf._viterbi(observation_sequence=x)
            out[f"u{i}_{tag}_score"], out[f"u{i}_{tag}_path"] = np.float32(score), path
            out[f"u{i}_{tag}_labels"] = np.array(inf.predict(x))
    out["n_utts"] = np.int64(len(utts))
    np.savez_compressed(os.path.join(OUT, "hmm_g7_wide.npz"), **out)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7"]      # every set has its own seed
    if "g1" in which:
        g1_emission(np.random.default_rng(101))
    if "g2" in which:
        g2_single(np.random.default_rng(102))
    if "g3" in which:
        g3_loop(np.random.default_rng(103))
    if "g4" in which:
        g4_labels()
    if "g5" in which:
        g5_training(np.random.default_rng(105))
    if "g6" in which:
        g6_collection(np.random.default_rng(106))
    if "g7" in which:
        g7_wide(np.random.default_rng(107))
    print("wrote", sorted(f for f in os.listdir(OUT) if f.startswith("hmm_")))
