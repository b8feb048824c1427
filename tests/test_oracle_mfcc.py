"""Known-answer tests for the MFCC oracle.  The reference delegates this stage to
librosa, which is not available, so these checks pin the restatement to
closed-form answers rather than to reference output (parity unpinned)."""
import numpy as np
import pytest
import scipy.fftpack

from oracle import mfcc_oracle as M


def test_frame_count():
    for n in (1440, 1599, 1600, 32000, 32159):
        y = np.zeros(n, np.float32)
        assert M.stft_power(y).shape == (161, 1 + n // 160)


def test_mel_edges_are_integer_mels():
    # fmin/fmax of the reference are mel 2 and mel 43 on the Slaney scale: 42 edges one mel apart
    edges = np.linspace(M.hz_to_mel_slaney(M.FMIN), M.hz_to_mel_slaney(M.FMAX), 42)
    assert np.allclose(edges, np.arange(2, 44), atol=2e-4)
    hz = M.mel_to_hz_slaney(edges)
    assert np.allclose(np.diff(hz[:14]), 200.0 / 3, rtol=1e-3)          # linear part below 1 kHz


def test_mel_filterbank_shape_and_norm():
    w = M.mel_filterbank()
    assert w.shape == (40, 161) and w.dtype == np.float32
    assert np.all(w >= 0)
    assert np.all(w[:, 0] == 0) and np.all(w[:, -20:].sum(axis=0)[-15:] == 0)   # nothing below fmin / above fmax
    # Slaney normalisation: every triangle has (continuous) area 1 in Hz; sampled at 50 Hz bins
    # the Riemann sum 50 * sum(w) is close to 1 for the wide upper bands
    assert np.allclose(50.0 * w[25:].sum(axis=1), 1.0, atol=0.06)
    assert len({int(np.argmax(r)) for r in w}) == 40                    # distinct peak bins


def test_bin_centred_sinusoid_power_and_band():
    sr, k, amp = 16000, 40, 1000.0                                       # 2 kHz, exactly bin 40
    n = np.arange(16000)
    y = (amp * np.cos(2 * np.pi * (50.0 * k) * n / sr)).astype(np.float32)
    p = M.stft_power(y)
    mid = p[:, 50]
    assert np.argmax(mid) == k
    assert mid[k] == pytest.approx((amp * 160 / 2) ** 2, rel=1e-4)       # sum(hann)=160, half in each sideband
    assert mid[k - 1] == pytest.approx(mid[k] / 4, rel=1e-3)             # Hann main lobe: -6 dB neighbours
    assert mid[k + 3] < 1e-8 * mid[k]
    mel = M.mel_power(y)
    w = M.mel_filterbank()
    assert np.argmax(mel[:, 50]) == np.argmax(w[:, k])


def test_power_to_db_range_and_silence():
    rng = np.random.default_rng(0)
    s = (rng.uniform(0, 1, (40, 30)) ** 8 * 1e6).astype(np.float32)
    db = M.power_to_db(s)
    assert db.dtype == np.float32 and db.max() == 0.0 and db.min() >= -80.0
    assert np.all(M.power_to_db(np.zeros((40, 12), np.float32)) == 0.0)
    feats = M.mfcc_features(np.zeros(3200, np.float32))
    assert feats.shape == (39, 21) and np.all(feats == 0.0)


def test_dct_matches_closed_form():
    rng = np.random.default_rng(1)
    L = rng.standard_normal((40, 7)).astype(np.float32) * 30
    got = scipy.fftpack.dct(L, axis=0, type=2, norm="ortho")[:13]
    n = np.arange(40)
    k = np.arange(13)[:, None]
    basis = np.cos(np.pi * k * (2 * n + 1) / 80.0) * np.sqrt(2.0 / 40)
    basis[0] *= np.sqrt(0.5)
    assert np.allclose(got, basis @ L.astype(np.float64), rtol=0, atol=2e-4)
    assert np.allclose(basis @ basis.T, np.eye(13), atol=1e-12)


def test_delta_closed_forms():
    T = 30
    t = np.arange(T, dtype=np.float64)
    ramp = (3.0 * t + 1.0)[None, :].astype(np.float32)
    assert np.allclose(M.delta(ramp, 1), 3.0, atol=1e-5)                 # edges included
    quad = (0.5 * t * t - 2 * t)[None, :].astype(np.float32)
    assert np.allclose(M.delta(quad, 2), 1.0, atol=1e-4)
    rng = np.random.default_rng(2)
    c = rng.standard_normal((13, T)).astype(np.float32)
    d1, d2 = M.delta(c, 1), M.delta(c, 2)
    k = np.arange(-4, 5)
    c2 = np.array([28, 7, -8, -17, -20, -17, -8, 7, 28]) / 462.0
    for tt in range(4, T - 4):
        assert np.allclose(d1[:, tt], (c[:, tt - 4:tt + 5] * k).sum(axis=1) / 60.0, atol=1e-6)
        assert np.allclose(d2[:, tt], (c[:, tt - 4:tt + 5] * c2).sum(axis=1), atol=1e-6)
    # the polynomial edge fit makes the first/last four frames repeat the nearest interior value
    for tt in range(4):
        assert np.allclose(d1[:, tt], d1[:, 4], atol=1e-6) and np.allclose(d2[:, tt], d2[:, 4], atol=1e-6)
        assert np.allclose(d1[:, T - 1 - tt], d1[:, T - 5], atol=1e-6)
        assert np.allclose(d2[:, T - 1 - tt], d2[:, T - 5], atol=1e-6)
    with pytest.raises(ValueError):
        M.delta(c[:, :8], 1)


def test_normalisation_is_per_frame():
    rng = np.random.default_rng(3)
    y = (rng.standard_normal(8000) * 500).astype(np.float32)
    f = M.mfcc_features(y)
    assert f.shape == (39, 51) and f.dtype == np.float32
    assert np.allclose(f[:13].mean(axis=0), 0, atol=1e-5)
    assert np.allclose(f[:13].std(axis=0), 1, atol=1e-4)
    g = M.mfcc_features((2 * y).astype(np.float32))                       # ref=max makes the chain gain invariant
    assert np.allclose(f, g, atol=2e-3)
    assert M.mfcc_batch([y], 16000)[0].shape == (51, 39)


def test_input_validation():
    with pytest.raises(TypeError):
        M.mfcc_features([0.0] * 2000)
    with pytest.raises(ValueError):
        M.mfcc_features(np.zeros((2, 2000), np.float32))


# ---- what the reference's own mfcc.py defines, against outputs of the reference itself -----------------------
# tests/golden/mfcc_g8_glue.npz (tools/make_golden_mfcc_glue.py): the real MFCC.normalize_mfccs, and the real
# MFCC.__post_init__ / MFCC.batch run over a stand-in librosa whose stages are this oracle's.  Pins mfcc.py:39-43,
# 50-69, 84 - NOT librosa's arithmetic.
def _glue():
    import os

    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mfcc_g8_glue.npz"), allow_pickle=False)


def test_g8_normalize_mfccs_equals_the_reference():
    g = _glue()
    for i in range(5):
        got = M.normalize_statics(g[f"norm{i}_in"])
        assert got.dtype == g[f"norm{i}_out"].dtype
        assert np.array_equal(got, g[f"norm{i}_out"]), i
    # the normalisation runs across the 13 coefficients of one frame, not along time
    x = g["norm2_in"]
    assert np.allclose(g["norm2_out"].mean(axis=0), 0, atol=1e-5) and not np.allclose(g["norm2_out"].mean(axis=1), 0, atol=1e-2)
    assert np.array_equal(g["norm4_out"], np.zeros_like(x[:, :10]))            # zero variance: (x - mean) / 1e-8 = 0


def test_g8_feature_layout_equals_the_reference():
    import json

    g = _glue()
    calls = json.loads(str(g["calls_of_first_clip"]))
    assert [c[0] for c in calls] == ["melspectrogram", "power_to_db", "mfcc", "delta", "delta"]
    assert calls[3][1] == {"order": 1, "input_is_the_unnormalised_mfcc_output": True}
    assert calls[4][1] == {"order": 2, "input_is_the_unnormalised_mfcc_output": True}
    sigs = [g[f"sig{i}"] for i in range(int(g["n_signals"]))]
    batch = M.mfcc_batch(sigs, 16000)
    for i, s in enumerate(sigs):
        f = M.mfcc_features(s)
        assert f.dtype == g[f"feat{i}"].dtype and np.array_equal(f, g[f"feat{i}"]), i
        assert np.array_equal(batch[i], g[f"batch{i}"]) and batch[i].shape == (1 + s.shape[0] // 160, 39), i
        # rows 0..12 normalised statics, 13..25 delta, 26..38 delta-delta of the UN-normalised statics
        c = M.dct_mfcc(M.power_to_db(M.mel_power(s)))
        assert np.array_equal(g[f"feat{i}"][13:26], M.delta(c, 1)) and np.array_equal(g[f"feat{i}"][26:], M.delta(c, 2))
        assert not np.allclose(g[f"feat{i}"][13:26], M.delta(M.normalize_statics(c), 1), atol=1e-3)
