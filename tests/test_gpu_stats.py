"""lsr_accumulate_stats (counting sort by slot + float64 MFMA Gram matrices) against NumPy float64 sums."""
import numpy as np
import pytest

from loe_speech_recognition_amd import _native

pytestmark = pytest.mark.gpu


def numpy_stats(x, slots, n):
    x = x.astype(np.float64)
    count = np.zeros(n)
    sum_x = np.zeros((n, 39))
    sum_xx = np.zeros((n, 39, 39))
    for s in range(n):
        sel = x[slots == s]
        count[s] = sel.shape[0]
        sum_x[s] = sel.sum(axis=0)
        sum_xx[s] = sel.T @ sel
    return count, sum_x, sum_xx


def device_stats(x, slots, n):
    eng = _native.get_engine()
    c, sx, sxx = eng.accumulate_stats(eng.to_device(x), eng.to_device(slots.astype(np.int32)), x.shape[0], n)
    return c.cpu().numpy(), sx.cpu().numpy(), sxx.cpu().numpy()


def check(x, slots, n):
    c, sx, sxx = device_stats(x, slots, n)
    wc, wsx, wsxx = numpy_stats(x, slots, n)
    assert np.array_equal(c, wc)                                   # counts are exact
    scale_x = np.maximum(1.0, np.abs(x).astype(np.float64).sum(axis=0))
    assert np.max(np.abs(sx - wsx) / scale_x) < 1e-13
    scale = np.maximum(1.0, np.einsum("i,j->ij", scale_x, np.abs(x).max(axis=0).astype(np.float64)))
    assert np.max(np.abs(sxx - wsxx) / scale) < 1e-13
    assert np.array_equal(sxx, np.transpose(sxx, (0, 2, 1)))       # exactly symmetric
    return c, sx, sxx


@pytest.mark.parametrize("n_frames,n_slots", [(1, 1), (3, 5), (4, 2), (1023, 7), (1025, 58), (50000, 58), (20000, 300)])
def test_runs_of_slots(n_frames, n_slots):
    rng = np.random.default_rng(n_frames + n_slots)
    x = (rng.standard_normal((n_frames, 39)) * rng.uniform(0.1, 30, size=39)).astype(np.float32)
    # runs of equal slot like an alignment produces, dropped frames (-1) and slots that never occur
    slots = np.empty(n_frames, np.int64)
    f = 0
    while f < n_frames:
        run = int(rng.integers(1, 40))
        slots[f:f + run] = rng.integers(-1, max(1, n_slots - 2))
        f += run
    check(x, slots, n_slots)


def test_single_slot_and_all_dropped():
    rng = np.random.default_rng(3)
    x = rng.standard_normal((5000, 39)).astype(np.float32)
    check(x, np.zeros(5000, np.int64), 1)
    c, sx, sxx = check(x, np.full(5000, -1, np.int64), 4)
    assert not c.any() and not sx.any() and not sxx.any()
    check(x, np.full(5000, 9, np.int64), 4)            # out-of-range slots are ignored like -1


def test_bitwise_reproducible():
    rng = np.random.default_rng(4)
    x = rng.standard_normal((30000, 39)).astype(np.float32)
    slots = rng.integers(-1, 58, size=30000)
    a = device_stats(x, slots, 58)
    b = device_stats(x, slots, 58)
    for u, v in zip(a, b):
        assert np.array_equal(u, v)


def numpy_remux(paths, frame_off, chain_off, chain_state, chain_word, n_states, drop_last_run):
    paths = paths.astype(np.int64)
    slots = np.full(paths.shape[0], -1, np.int32)
    trans = np.zeros((n_states, n_states), np.int64)
    for b in range(frame_off.shape[0] - 1):
        lo, hi = int(frame_off[b]), int(frame_off[b + 1])
        c0 = int(chain_off[b])
        w_last = chain_word[c0 + paths[hi - 1]]
        for f in range(lo, hi):
            g, w = chain_state[c0 + paths[f]], chain_word[c0 + paths[f]]
            keep = not drop_last_run or w != w_last
            if keep:
                slots[f] = g
            if keep and f + 1 < hi:
                g1, w1 = chain_state[c0 + paths[f + 1]], chain_word[c0 + paths[f + 1]]
                if (not drop_last_run or w1 != w_last) and w1 == w:
                    trans[g, g1] += 1
    return slots, trans


@pytest.mark.parametrize("n_states,drop", [(5, False), (58, True), (128, True), (131, True)])
def test_remux_slots_and_transition_counts(n_states, drop):
    """Path -> statistics slot and transition counts (lsr_remux_slots) against a per-frame Python loop; 128 states is
    the largest transition table counted in LDS, 131 takes the global-atomic path."""
    eng = _native.get_engine()
    rng = np.random.default_rng(n_states)
    word_size = 5 if n_states >= 5 else n_states
    n_words_model = n_states // word_size
    B = 300
    lens = rng.integers(1, 120, size=B)
    frame_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    chain_off, chain_state, chain_word, paths = [0], [], [], []
    for b in range(B):
        n_w = int(rng.integers(1, 9))
        for k in range(n_w):
            w = int(rng.integers(n_words_model))
            chain_state.extend(range(w * word_size, (w + 1) * word_size))
            chain_word.extend([k] * word_size)
        n_chain = n_w * word_size
        chain_off.append(chain_off[-1] + n_chain)
        steps = rng.integers(0, 3, size=lens[b])                  # a left-to-right walk with skips
        steps[0] = 0
        paths.append(np.minimum(np.cumsum(steps), n_chain - 1))
    paths = np.concatenate(paths).astype(np.int8)
    chain_off = np.asarray(chain_off, np.int64)
    chain_state = np.asarray(chain_state, np.int32)
    chain_word = np.asarray(chain_word, np.int32)
    want_slots, want_trans = numpy_remux(paths, frame_off, chain_off, chain_state, chain_word, n_states, drop)
    t = eng.torch
    slots = eng.empty((paths.shape[0],), t.int32)
    trans = eng.empty((n_states, n_states), t.int64)
    # the device copies are held in variables for the duration of the calls
    d_paths, d_frame_off, d_chain_off = eng.to_device(paths), eng.to_device(frame_off), eng.to_device(chain_off)
    d_state, d_word = eng.to_device(chain_state), eng.to_device(chain_word)
    for _ in range(2):                                            # twice: the table is cleared per call
        eng._check(eng.lib.lsr_remux_slots(eng.ctx, eng._ptr(d_paths), eng._ptr(d_frame_off), eng._ptr(d_chain_off),
                                           eng._ptr(d_state), eng._ptr(d_word), B, paths.shape[0], n_states,
                                           1 if drop else 0, eng._ptr(slots), eng._ptr(trans)))
        eng.synchronize()
    assert np.array_equal(slots.cpu().numpy(), want_slots)
    assert np.array_equal(trans.cpu().numpy(), want_trans)
    assert want_trans.sum() > 1000


def test_full_size_training_iteration_properties():
    """Config 4 at its full size (12 000 clips of 1-7 digits, 2.5 M frames) through size-independent properties:
    (i) forced-alignment paths are the same bits whether the emission pass scores every state or only the states
    the chains can reach; (ii) the statistics of the whole corpus equal the sum of the statistics of two contiguous
    shards - what the all-reduce of a two-rank run adds up - exactly for the counts and to float64 rounding for the
    sums; (iii) the M-step from the summed shards gives the models of the single-process iteration."""
    import multiprocessing as mp

    from loe_speech_recognition_amd import HiddenMarkovModelTrainContinuous, _native, synthetic

    with mp.get_context("fork").Pool(8) as pool:
        parts = pool.starmap(synthetic.batch, [(4, 400, [1, 2, 3, 4, 5, 7], 400 * k) for k in range(30)])
    sigs = [s for p in parts for s in p[0]]
    truth = [d for p in parts for d in p[1]]
    assert len(sigs) == 12000
    eng = _native.get_engine()
    t = eng.torch
    fb = eng.mfcc(eng.audio_batch(sigs))
    models = synthetic.load_word_models()

    def trainer():
        tc = HiddenMarkovModelTrainContinuous(isTqdm=False)
        for lab in synthetic.WORDS + ["S"]:
            tc.add_model(models[lab])
        return tc

    def shard(a, b):
        f0, f1 = int(fb.host_off[a]), int(fb.host_off[b])
        off = eng.to_device((fb.host_off[a:b + 1] - fb.host_off[a]).astype(np.int64))
        return _native.FeatureBatch(fb.feats[f0:f1], off, fb.lengths[a:b])

    # (i) both emission forms, full size
    whole = trainer()
    whole.use_device_batch(fb, truth)
    paths = {}
    for mode in ("all", "reachable"):
        eng.chain_emission_mode(mode)
        try:
            paths[mode] = whole._align(fb, truth)[0].clone()
        finally:
            eng.chain_emission_mode("auto")
    assert t.equal(paths["all"], paths["reachable"])
    # (ii) whole corpus against the sum of two shards
    stats, n_states = whole._local_statistics(None)
    stats = stats.cpu().numpy()
    total = np.zeros_like(stats)
    for a, b in ((0, 5000), (5000, 12000)):
        part = trainer()
        part.use_device_batch(shard(a, b), truth[a:b])
        total += part._local_statistics(None)[0].cpu().numpy()
    n_int = n_states                                               # counts, then (after the sums) transition counts
    assert np.array_equal(stats[:n_int], total[:n_int]) and stats[:n_int].sum() > 2.0e6
    assert np.array_equal(stats[-n_states * n_states:], total[-n_states * n_states:])
    assert np.allclose(stats, total, rtol=1e-12, atol=1e-6)
    # (iii) the M-step of the summed shards against the single-process iteration
    d = 39
    base, _ = whole._layout()
    summed = trainer()
    summed._m_steps(base, total[:n_states], total[n_states:n_states * (1 + d)].reshape(n_states, d),
                    total[n_states * (1 + d):n_states * (1 + d + d * d)].reshape(n_states, d, d),
                    total[-n_states * n_states:].reshape(n_states, n_states))
    whole._iteration(None)
    for lab in whole._models_loaded:
        a, b = whole._trainable_models[lab], summed._trainable_models[lab]
        assert np.allclose(a._means, b._means, rtol=1e-5, atol=1e-6) and np.allclose(a._covariances, b._covariances, rtol=1e-4, atol=1e-6)
        assert np.array_equal(a._transition_probs.to_array(), b._transition_probs.to_array(), equal_nan=True)
