"""Forced-alignment decoding with emission scores of the reachable states only (lsr_chain_emission_mode):
scores and paths must carry the bits of the all-states form, and both must equal the oracle's chain
recursion (hidden_markov_model.py:160-208 on the stacked model of hidden_markov_model.py:638-664) run on
the device's own scores."""
import numpy as np
import pytest

from oracle import hmm_oracle as O

pytestmark = pytest.mark.gpu


def _model(rng, n_words, sizes):
    """Random word models: (means, whiten-able covariances, band per word)."""
    from loe_speech_recognition_amd import MultivariateNormal

    gaussians, bands, base = [], [], []
    n = 0
    for w in range(n_words):
        k = int(sizes[w])
        base.append(n)
        n += k
        for _ in range(k):
            a = rng.standard_normal((39, 39))
            cov = (a @ a.T / 39 + 1e-2 * np.eye(39)).astype(np.float32)
            gaussians.append(MultivariateNormal.from_means_covariances(rng.standard_normal(39).astype(np.float32), cov))
        with np.errstate(divide="ignore"):
            tr = np.log(rng.dirichlet(np.ones(3), size=k)).astype(np.float32)
        band = np.full((3, k), -np.inf, np.float32)
        for s in range(k):
            for d in range(3):
                if s - d >= 0:
                    band[d, s] = tr[s, d]
        bands.append(band)
    return gaussians, bands, np.array(base), n


def _chains(rng, n_utts, n_words, sizes, base, bands, max_states):
    off, ids, band, start = [0], [], [], []
    for _ in range(n_utts):
        states = 0
        first = True
        while True:
            w = int(rng.integers(n_words))
            if states + sizes[w] > max_states or (not first and rng.random() < 0.25):
                break
            ids.append(base[w] + np.arange(sizes[w]))
            b = bands[w].copy()
            st = np.zeros(sizes[w], np.uint8)
            if first:
                st[0] = 1
            else:                        # keys that cross a word boundary read 0.0 (transition_probability.py:20-23)
                b[1, 0] = 0.0
                b[2, :2] = 0.0
            band.append(b)
            start.append(st)
            states += sizes[w]
            first = False
        off.append(off[-1] + states)
    return (np.array(off, np.int64), np.concatenate(ids).astype(np.int32), np.concatenate(band, axis=1),
            np.concatenate(start))


@pytest.mark.parametrize("seed,n_words,max_states", [(0, 12, 40), (1, 25, 64), (2, 30, 127), (3, 6, 20)])
def test_reachable_states_only_gives_the_same_bits(seed, n_words, max_states):
    from loe_speech_recognition_amd import _native
    from loe_speech_recognition_amd.hidden_markov_model import stack_gaussians

    rng = np.random.default_rng(4100 + seed)
    sizes = rng.integers(2, 6, size=n_words)
    gaussians, bands, base, n_states = _model(rng, n_words, sizes)
    eng = _native.get_engine()
    eng.set_model(None, *stack_gaussians(gaussians))
    n_utts = 90
    off, ids, band, start = _chains(rng, n_utts, n_words, sizes, base, bands, max_states)
    # lengths from 1 frame (several utterances inside one 64-frame tile) to a few tiles
    lens = rng.integers(1, 230, size=n_utts)
    lens[:6] = [1, 2, 9, 64, 63, 65]
    xs = [(1.5 * rng.standard_normal((int(t), 39))).astype(np.float32) for t in lens]
    fb = eng.feature_batch(xs)
    got = {}
    for mode in ("all", "reachable", "auto"):
        eng.chain_emission_mode(mode)
        try:
            scores, paths, final = eng.decode_chain(fb, off, ids, band, start, want_paths=True,
                                                    want_final=int(np.diff(off).max()) <= 64)
            got[mode] = (scores.cpu().numpy(), paths.cpu().numpy(), None if final is None else final.cpu().numpy())
        finally:
            eng.chain_emission_mode("auto")
    for mode in ("reachable", "auto"):
        assert np.array_equal(got["all"][0], got[mode][0], equal_nan=True), mode
        assert np.array_equal(got["all"][1], got[mode][1]), mode
        if got["all"][2] is not None:
            lens_c = np.diff(off)
            for b in range(n_utts):          # entries of chain positions (columns beyond the chain are unspecified)
                assert np.array_equal(got["all"][2][b, :lens_c[b]], got[mode][2][b, :lens_c[b]], equal_nan=True), (mode, b)
    # against the oracle's recursion on the device's scores
    ll = eng.emission(fb).cpu().numpy()
    t0 = 0
    for b, x in enumerate(xs):
        sl = slice(off[b], off[b + 1])
        emis = ll[t0:t0 + x.shape[0]][:, ids[sl]]
        score, path, _ = O.viterbi_chain_fast(emis, band[:, sl])
        a = got["reachable"][0][b]
        assert a == score or (np.isnan(a) and np.isnan(score)), (b, a, score)
        assert np.array_equal(got["reachable"][1][t0:t0 + x.shape[0]], path), b
        t0 += x.shape[0]


def test_reachable_mode_scores_fewer_pairs_and_leaves_other_columns():
    """The masked pass writes the columns of the chain's states and nothing else: poison the score buffer through
    a full pass of another model, then check that a one-word chain over a 12-word model decodes as before."""
    from loe_speech_recognition_amd import _native
    from loe_speech_recognition_amd.hidden_markov_model import stack_gaussians

    rng = np.random.default_rng(99)
    sizes = np.full(12, 5)
    gaussians, bands, base, n_states = _model(rng, 12, sizes)
    eng = _native.get_engine()
    eng.set_model(None, *stack_gaussians(gaussians))
    xs = [(1.5 * rng.standard_normal((int(t), 39))).astype(np.float32) for t in (100, 37, 300)]
    fb = eng.feature_batch(xs)
    off = np.array([0, 5, 10, 15], np.int64)
    ids = np.concatenate([base[3] + np.arange(5), base[11] + np.arange(5), base[0] + np.arange(5)]).astype(np.int32)
    band = np.concatenate([bands[3], bands[11], bands[0]], axis=1)
    start = np.tile(np.array([1, 0, 0, 0, 0], np.uint8), 3)
    out = {}
    for mode in ("all", "reachable"):
        eng.chain_emission_mode(mode)
        try:
            s, p, _ = eng.decode_chain(fb, off, ids, band, start)
            out[mode] = (s.cpu().numpy(), p.cpu().numpy())
        finally:
            eng.chain_emission_mode("auto")
    assert np.array_equal(out["all"][0], out["reachable"][0]) and np.array_equal(out["all"][1], out["reachable"][1])
    assert np.isfinite(out["all"][0]).all()
