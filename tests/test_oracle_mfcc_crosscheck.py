"""The MFCC oracle against a second, independently written restatement of the librosa chain.

librosa itself is not installed, so the oracle stays unpinned against the library the reference calls
(mfcc.py:31-43).  `transformers.audio_utils` (installed here) was written by other authors to reproduce
librosa's mel filterbank, STFT power spectrogram and power_to_db; agreement with it up to the log-mel stage
rules out a misreading of the published algorithm on our side (Slaney scale and normalisation, periodic Hann,
frame count and centring, amin / top_db), but not the choice of padding mode, which is an argument here too.
The stages after the log-mel spectrogram are SciPy calls (dct, savgol_filter) in both librosa and the oracle."""
import os
import sys
import warnings

import numpy as np
import pytest
import scipy.signal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import mfcc_oracle as M  # noqa: E402

with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    A = pytest.importorskip("transformers.audio_utils")


def _clips():
    rng = np.random.default_rng(11)
    t = np.arange(16000) / 16000.0
    yield (0.1 * rng.standard_normal(16000) * np.sin(2 * np.pi * 1.7 * t)).astype(np.float32)
    yield (0.5 * np.sin(2 * np.pi * 440.0 * t) + 0.01 * rng.standard_normal(16000)).astype(np.float32)[:12345]
    yield (3000.0 * rng.standard_normal(4000)).astype(np.float32)                 # int16-scale amplitudes
    yield np.concatenate([np.zeros(2000, np.float32), 0.2 * rng.standard_normal(3000).astype(np.float32)])


def test_mel_filterbank_and_window_agree():
    fb = A.mel_filter_bank(M.N_BINS, M.N_MELS, M.FMIN, M.FMAX, 16000, norm="slaney", mel_scale="slaney")
    ours = M.mel_filterbank()
    assert fb.shape == (M.N_BINS, M.N_MELS)
    assert np.array_equal(fb.T > 0, ours > 0)                                  # same support of every band
    assert np.abs(fb.T - ours).max() < 2e-9                                    # float32 rounding of weights <= 0.015
    w = A.window_function(M.N_FFT, "hann", periodic=True)
    assert np.abs(w - scipy.signal.get_window("hann", M.N_FFT, fftbins=True)).max() < 1e-15


def test_log_mel_spectrogram_agrees():
    fb = A.mel_filter_bank(M.N_BINS, M.N_MELS, M.FMIN, M.FMAX, 16000, norm="slaney", mel_scale="slaney")
    w = A.window_function(M.N_FFT, "hann", periodic=True)
    for y in _clips():
        S = A.spectrogram(y, w, M.N_FFT, M.HOP, fft_length=M.N_FFT, power=2.0, center=True, pad_mode="constant",
                          mel_filters=fb, mel_floor=1e-10)
        assert S.shape == (M.N_MELS, M.num_frames(y.shape[0]))
        db = A.power_to_db(S, reference=float(S.max()), min_value=1e-10, db_range=80.0)
        ours = M.power_to_db(M.mel_power(y))
        # float64 there, float32 spectrum here: 1e-4 dB is 2e-5 relative on the mel energies
        assert np.abs(db - ours).max() < 1e-4
        # with reflect padding only the frames that reach over a clip edge differ
        Sr = A.spectrogram(y, w, M.N_FFT, M.HOP, fft_length=M.N_FFT, power=2.0, center=True, pad_mode="reflect",
                           mel_filters=fb, mel_floor=1e-10)
        assert np.allclose(Sr[:, 1:-1], S[:, 1:-1], rtol=1e-12, atol=0)
