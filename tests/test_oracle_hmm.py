"""The CPU oracle (oracle/hmm_oracle.py) against fixtures computed by the real
reference code (tools/make_golden_hmm.py).  Scores and paths must be identical."""
import os

import numpy as np
import pytest

from oracle import hmm_oracle as O

LOOP_LABELS = ["1", "2", "3", "4", "5", "6", "7", "8", "9", "O", "S", "Z"]
BETWEEN = {"m100": -100, "m250": -250, "dflt": np.log(0.005), "m3f": -3.0}


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def word(g, prefix):
    return O.WordModelRef(prefix, g[f"{prefix}_means"], g[f"{prefix}_cov"], g[f"{prefix}_trans"])


def test_g1_emission_ref_bit_exact(golden_dir):
    g = load(golden_dir, "hmm_g1_emission.npz")
    gauss = [O.GaussianRef(m, c) for m, c in zip(g["means"], g["cov"])]
    x = g["x"]
    ll = O.emission_ref(x[:8], gauss)
    assert np.array_equal(ll, g["ll"][:8])
    mu, U, ld = O.export_gaussians(gauss)
    assert np.array_equal(U[:4], g["U_first4"])
    assert np.array_equal(ld, g["logdet"])
    fast = O.emission_fast(x, mu, U, ld)
    # BLAS order differs between the batched and the per-call product: allow one float32 ulp
    assert np.all(np.abs(fast - g["ll"]) <= np.spacing(np.abs(g["ll"])))
    assert np.mean(fast != g["ll"]) < 1e-3


def test_singular_covariance_is_rejected():
    cov = np.eye(39, dtype=np.float32)
    cov[3, 3] = 0.0
    with pytest.raises(np.linalg.LinAlgError):
        O.GaussianRef(np.zeros(39, np.float32), cov)


@pytest.mark.parametrize("n", [3, 5])
def test_g2_single_model(golden_dir, n):
    g = load(golden_dir, "hmm_g2_single.npz")
    m = O.WordModelRef("w", g[f"n{n}_means"], g[f"n{n}_cov"], g[f"n{n}_trans"])
    for tag in ("T9", "T50", "T200", "late"):
        x = g[f"n{n}_{tag}_x"]
        if tag in ("T9", "late"):
            score, path = m.align(x)                 # reference loop structure
            assert score == g[f"n{n}_{tag}_score"] and np.array_equal(path, g[f"n{n}_{tag}_path"])
            assert path.dtype == np.int8 and isinstance(score, np.float32)
        score, path = m.align_fast(x)
        assert score == g[f"n{n}_{tag}_score"]
        assert np.array_equal(path, g[f"n{n}_{tag}_path"])
    # the terminal state entered on the last frame never shows in the path
    assert g[f"n{n}_late_path"][-1] == g[f"n{n}_late_path"][-2]


@pytest.mark.parametrize("tag", ["n35", "n59"])
def test_g2_alignment_chain(golden_dir, tag):
    g = load(golden_dir, "hmm_g2_single.npz")
    words = {lab: word(g, f"chain_{lab}") for lab in "123S"}
    for lab in words:
        words[lab].label = lab
    chain = str(g[f"{tag}_chain"])
    ltp, gauss, bounds = O.stack_models([words[c] for c in chain])
    assert ltp.get(2, 3) == 0.0 and ltp.get(1, 3) == 0.0     # cross-word keys are missing -> free
    mu, U, ld = O.export_gaussians(gauss)
    band = O.band_from_ltp(ltp, len(gauss))
    for T in (50, 200):
        key = f"{tag}_T{T}_x"
        if key not in g.files:
            continue
        x = g[key]
        score, path, _ = O.viterbi_chain_fast(O.emission_fast(x, mu, U, ld), band)
        assert score == g[f"{tag}_T{T}_score"]
        assert np.array_equal(path, g[f"{tag}_T{T}_path"])
    x = g[f"{tag}_T200_x"][:40] if tag == "n35" else g[f"{tag}_T200_x"][:70]
    s_ref, p_ref = O.viterbi_chain_ref(x, ltp, gauss)
    s_fast, p_fast, _ = O.viterbi_chain_fast(O.emission_fast(x, mu, U, ld), band)
    assert s_ref == s_fast and np.array_equal(p_ref, p_fast)


@pytest.mark.parametrize("tag", ["n75", "n123"])
def test_g7_long_alignment_chain(golden_dir, tag):
    """Chains beyond 64 states (reference fixture): the restatement has no size limit of its own."""
    g = load(golden_dir, "hmm_g7_wide.npz")
    words = {lab: word(g, f"chain_{lab}") for lab in "123S"}
    ltp, gauss, _ = O.stack_models([words[c] for c in str(g[f"{tag}_chain"])])
    assert len(gauss) == int(tag[1:])
    mu, U, ld = O.export_gaussians(gauss)
    band = O.band_from_ltp(ltp, len(gauss))
    for T in (160, 300):
        score, path, _ = O.viterbi_chain_fast(O.emission_fast(g[f"{tag}_T{T}_x"], mu, U, ld), band)
        assert score == g[f"{tag}_T{T}_score"] and np.array_equal(path, g[f"{tag}_T{T}_path"])


def test_g7_loop_of_123_states(golden_dir):
    g = load(golden_dir, "hmm_g7_wide.npz")
    order = [str(x) for x in g["loop_order"]]
    words = [word(g, f"loop_{lab}") for lab in order]
    for w, lab in zip(words, order):
        w.label = lab
    ltp, gauss, bounds = O.stack_models(words)
    assert bounds.num_states == 123
    mu, U, ld = O.export_gaussians(gauss)
    band = O.band_from_ltp(ltp, 123, [bounds.find_lower(s) for s in range(123)])
    for i in range(int(g["n_utts"])):
        emis = O.emission_fast(g[f"u{i}_x"], mu, U, ld)
        for tag in ("m100", "dflt", "m3f"):
            score, path = O.viterbi_loop_fast(emis, band, bounds.lower, bounds.upper, BETWEEN[tag])
            assert score == g[f"u{i}_{tag}_score"], (i, tag)
            assert np.array_equal(path, g[f"u{i}_{tag}_path"]), (i, tag)
            assert "".join(bounds.get_labels(path)) == str(g[f"u{i}_{tag}_labels"])


def loop_setup(g):
    words = [word(g, lab) for lab in LOOP_LABELS]
    ltp, gauss, bounds = O.stack_models(words)
    assert bounds.lower == [0, 5, 10, 15, 20, 25, 30, 35, 40, 45, 50, 53]
    assert bounds.upper == [4, 9, 14, 19, 24, 29, 34, 39, 44, 49, 52, 57]
    return ltp, gauss, bounds


def test_g3_loop_decode_fast(golden_dir):
    g = load(golden_dir, "hmm_g3_loop.npz")
    ltp, gauss, bounds = loop_setup(g)
    mu, U, ld = O.export_gaussians(gauss)
    lower_of = [bounds.find_lower(s) for s in range(bounds.num_states)]
    band = O.band_from_ltp(ltp, bounds.num_states, lower_of)
    for i in range(int(g["n_utts"])):
        x = g[f"u{i}_x"]
        emis = O.emission_fast(x, mu, U, ld)
        for tag, between in BETWEEN.items():
            score, path = O.viterbi_loop_fast(emis, band, bounds.lower, bounds.upper, between)
            assert score == g[f"u{i}_{tag}_score"], (i, tag)
            assert np.array_equal(path, g[f"u{i}_{tag}_path"]), (i, tag)
            assert "".join(bounds.get_labels(path)) == str(g[f"u{i}_{tag}_labels"])


def test_g3_loop_decode_reference_structure(golden_dir):
    g = load(golden_dir, "hmm_g3_loop.npz")
    ltp, gauss, bounds = loop_setup(g)
    for i, tag in ((0, "m100"), (2, "dflt"), (3, "m3f")):
        x = g[f"u{i}_x"]
        score, path = O.viterbi_loop_ref(x, ltp, gauss, bounds, BETWEEN[tag])
        assert score == g[f"u{i}_{tag}_score"]
        assert np.array_equal(path, g[f"u{i}_{tag}_path"])


def test_g4_labels(golden_dir):
    g = load(golden_dir, "hmm_g4_labels.npz")
    bounds = O.BoundariesRef([3 if lab == "S" else 5 for lab in LOOP_LABELS], LOOP_LABELS)
    for i in range(int(g["n"])):
        assert "".join(bounds.get_labels(g[f"p{i}"])) == str(g[f"l{i}"])
        assert "".join(bounds.get_labels(g[f"p{i}"], skip_silence=False)) == str(g[f"k{i}"])


def trans_equal(a, b):
    return np.array_equal(np.nan_to_num(a, nan=-7.0), np.nan_to_num(b, nan=-7.0))


def test_g5_isolated_train_step(golden_dir):
    g = load(golden_dir, "hmm_g5_training.npz")
    m = O.WordModelRef("w", g["a_means0"], g["a_cov0"], g["a_trans0"])
    utts = [g[f"a_x{i}"] for i in range(int(g["a_n"]))]
    m.reestimate([(x, m.align_fast(x)[1]) for x in utts])
    assert np.array_equal(m.means, g["a_means1"])
    assert np.array_equal(m.cov, g["a_cov1"])
    assert trans_equal(m.trans, g["a_trans1"])


def test_g5_from_data(golden_dir):
    g = load(golden_dir, "hmm_g5_training.npz")
    utts = [g[f"b_x{i}"] for i in range(int(g["b_n"]))]
    m = O.WordModelRef.from_data("w", utts, n_states=5, max_iterations=6)
    assert np.array_equal(m.means, g["b_means"])
    assert np.array_equal(m.cov, g["b_cov"])
    assert trans_equal(m.trans, g["b_trans"])


def test_g5_continuous_iteration(golden_dir):
    g = load(golden_dir, "hmm_g5_training.npz")
    models = {lab: O.WordModelRef(lab, g[f"c_{lab}_means0"], g[f"c_{lab}_cov0"], g[f"c_{lab}_trans0"])
              for lab in "12OS"}
    labeled = {}
    for digits in [str(k) for k in g["c_keys"]]:
        labeled[digits] = [g[f"c_x_{digits}_{i}"] for i in range(3)]
    pieces, _ = O.continuous_iteration(models, labeled)
    for lab in "12OS":
        assert len(pieces[lab]) == int(g[f"c_{lab}_nseg"])
        assert np.array_equal(np.array([f.shape[0] for _, f, _ in pieces[lab]]), g[f"c_{lab}_seg_frames"])
        assert np.array_equal(np.concatenate([p for _, _, p in pieces[lab]]), g[f"c_{lab}_seg_paths"])
        assert np.array_equal(models[lab].means, g[f"c_{lab}_means1"])
        assert np.array_equal(models[lab].cov, g[f"c_{lab}_cov1"])
        assert trans_equal(models[lab].trans, g[f"c_{lab}_trans1"])


def test_g6_isolated_classifier_tie_order(golden_dir):
    g = load(golden_dir, "hmm_g6_collection.npz")
    labels = ["1", "2", "3", "4", "5", "6", "7", "8", "9", "O", "Z"]
    models = [word(g, lab) for lab in labels]
    preds = [str(p) for p in g["preds"]]
    assert preds[0] == "2" and preds[1] == "2"              # "8" is a copy of "2": the earlier label wins
    for i in range(5):
        x = g[f"x{i}"]
        assert O.classify_isolated(models, x) == preds[i]
        scores = np.array([m.align_fast(x)[0] for m in models], dtype=np.float32)
        assert np.array_equal(scores, g[f"scores{i}"])


def _random_gmm(rng, n_states, n_mix):
    weights = rng.dirichlet(np.ones(n_mix) * 2, size=n_states)
    comps = []
    for _ in range(n_states):
        row = []
        for _ in range(n_mix):
            a = rng.standard_normal((39, 39))
            row.append(O.GaussianRef((rng.standard_normal(39) * 2).astype(np.float32),
                                     (a @ a.T / 39 + 1e-2 * np.eye(39)).astype(np.float32)))
        comps.append(row)
    return weights, comps


def _g9(tag, golden_dir):
    g = np.load(os.path.join(golden_dir, "gmm_g9_emission.npz"), allow_pickle=False)
    comps = [[O.GaussianRef(m, c) for m, c in zip(ms, cs)] for ms, cs in zip(g[f"{tag}_means"], g[f"{tag}_cov"])]
    return g, comps


@pytest.mark.parametrize("tag", ["a", "b"])
def test_g9_mixture_emission_equals_the_reference(golden_dir, tag):
    """tests/golden/gmm_g9_emission.npz holds what the reference's own HMM_GMM.log_emission_prob returned
    (deprecated/gaussian_mixture_model.py:159-164, float64; tools/make_golden_gmm.py).  The loop restatement must
    reproduce those values exactly once rounded to float32 as the live code rounds, the vectorised one to an ulp."""
    g, comps = _g9(tag, golden_dir)
    assert "FileNotFoundError" in str(g["module_execution_stopped_at"])      # the class was taken from the real module
    want = g[f"{tag}_logprob"].astype(np.float32)
    got = O.emission_gmm_ref(g[f"{tag}_x"], g[f"{tag}_weights"], comps)
    assert np.array_equal(got, want)
    assert np.isfinite(want).all() and (g[f"{tag}_weights"][2, 0] == 0.0)   # the zero-weight component drops out
    fast = O.emission_gmm_fast(g[f"{tag}_x"], g[f"{tag}_weights"], comps)
    assert np.all(np.abs(fast - want) <= np.spacing(np.abs(want)))


def test_gmm_emission_restatements_agree():
    """Beyond the fixture: the restatement against the definition log sum_m w_m N_m and its vectorised form against
    the loop form on further random mixtures."""
    rng = np.random.default_rng(8)
    weights, comps = _random_gmm(rng, 3, 4)
    x = (rng.standard_normal((6, 39)) * 2).astype(np.float32)
    ref = O.emission_gmm_ref(x, weights, comps)
    from scipy.special import logsumexp

    direct = np.array([[logsumexp([np.log(weights[s][m]) + comps[s][m].frozen.logpdf(x[t]) for m in range(4)])
                        for s in range(3)] for t in range(6)])
    assert np.allclose(ref, direct, rtol=1e-6)
    fast = O.emission_gmm_fast(x, weights, comps)
    assert np.all(np.abs(fast - ref) <= np.spacing(np.abs(ref)))
    one = O.emission_gmm_ref(x, np.ones((3, 1)), [[c[0]] for c in comps])
    assert np.array_equal(one, O.emission_ref(x, [c[0] for c in comps]))
