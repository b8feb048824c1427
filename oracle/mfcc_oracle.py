"""CPU oracle for the MFCC front-end.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference (src/loe_speech_recognition/mfcc.py:31-43) does
all of its arithmetic inside librosa (melspectrogram -> power_to_db -> mfcc ->
delta x2), a third-party dependency that is neither vendored under
/root/reference nor installed here (declared without a version in the
reference's pyproject.toml:20; numpy 2.2.1 in requirements.txt implies
librosa >= 0.10.2).  The reference holds no test vectors for this stage either.
This file therefore restates librosa 0.10.x's published algorithm with the
NumPy/SciPy calls librosa itself makes (np.fft.rfft, scipy.fftpack.dct,
scipy.signal.savgol_filter) and is checked by known-answer tests
(tests/test_oracle_mfcc.py), not against librosa output.  A second restatement
of the same chain by other authors (transformers.audio_utils, installed here)
agrees with it up to the log-mel stage to 1e-5 dB
(tests/test_oracle_mfcc_crosscheck.py); that rules out a misreading of the
algorithm, it does not pin the padding mode or librosa's own rounding points.

What IS pinned is the part of the stage that the reference's own file defines: tests/golden/mfcc_g8_glue.npz
(tools/make_golden_mfcc_glue.py) holds outputs of the reference's real MFCC.normalize_mfccs, and of its real
MFCC.__post_init__ / MFCC.batch run with a stand-in ``librosa`` module whose four functions are the stage
functions below - that fixes mfcc.py:39-43,84 (both deltas taken from the un-normalised statics, the order
of the concatenation, the transposition) and mfcc.py:50-69, and nothing of librosa's arithmetic.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import it.
"""
from __future__ import annotations

from typing import List

import numpy as np
import scipy.fftpack
import scipy.signal

N_FFT = 320
HOP = 160
N_MELS = 40
N_BINS = N_FFT // 2 + 1
FMIN = 133.33
FMAX = 6855.4976
N_MFCC = 13
DELTA_WIDTH = 9


def num_frames(n_samples: int) -> int:
    """center=True pads n_fft//2 on both sides: T = 1 + N // hop."""
    return 1 + n_samples // HOP


def hz_to_mel_slaney(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    if f.ndim:
        big = f >= min_log_hz
        mels[big] = min_log_mel + np.log(f[big] / min_log_hz) / logstep
    elif f >= min_log_hz:
        mels = min_log_mel + np.log(f / min_log_hz) / logstep
    return mels


def mel_to_hz_slaney(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    if m.ndim:
        big = m >= min_log_mel
        freqs[big] = min_log_hz * np.exp(logstep * (m[big] - min_log_mel))
    elif m >= min_log_mel:
        freqs = min_log_hz * np.exp(logstep * (m - min_log_mel))
    return freqs


def mel_filterbank(sr: float = 16000, n_fft: int = N_FFT, n_mels: int = N_MELS,
                   fmin: float = FMIN, fmax: float = FMAX) -> np.ndarray:
    """librosa.filters.mel(htk=False, norm="slaney"): float32 [n_mels, 1 + n_fft//2]."""
    weights = np.zeros((n_mels, 1 + n_fft // 2), dtype=np.float32)
    fftfreqs = np.fft.rfftfreq(n=n_fft, d=1.0 / sr)
    mel_f = mel_to_hz_slaney(np.linspace(hz_to_mel_slaney(fmin), hz_to_mel_slaney(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2: n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis]
    return weights


def stft_power(y: np.ndarray) -> np.ndarray:
    """|STFT|**2 as float32 [161, T]: zero ("constant") centre padding, periodic
    Hann in float64, double-precision rFFT rounded to complex64, abs then square."""
    y = np.asarray(y, dtype=np.float32)
    ypad = np.pad(y, N_FFT // 2, mode="constant")
    T = num_frames(y.shape[0])
    idx = HOP * np.arange(T)[None, :] + np.arange(N_FFT)[:, None]
    frames = ypad[idx]                                           # [320, T] float32
    window = scipy.signal.get_window("hann", N_FFT, fftbins=True)[:, None]   # float64
    spec = np.fft.rfft(window * frames, axis=0).astype(np.complex64)
    return np.abs(spec) ** 2.0


def mel_power(y: np.ndarray, sr: float = 16000) -> np.ndarray:
    return np.dot(mel_filterbank(sr), stft_power(y))              # float32 [40, T]


def power_to_db(S: np.ndarray, amin: float = 1e-10, top_db: float = 80.0) -> np.ndarray:
    """librosa.power_to_db(S, ref=np.max)."""
    ref_value = np.max(S)
    log_spec = 10.0 * np.log10(np.maximum(amin, S))
    log_spec -= 10.0 * np.log10(np.maximum(amin, ref_value))
    return np.maximum(log_spec, log_spec.max() - top_db)


def delta(c: np.ndarray, order: int) -> np.ndarray:
    """librosa.feature.delta(width=9, mode="interp") is exactly this SciPy call."""
    if c.shape[-1] < DELTA_WIDTH:
        raise ValueError(f"width={DELTA_WIDTH} cannot exceed the number of frames {c.shape[-1]}")
    return scipy.signal.savgol_filter(c, DELTA_WIDTH, deriv=order, axis=-1, mode="interp", polyorder=order)


def dct_mfcc(log_mel: np.ndarray, n_mfcc: int = N_MFCC) -> np.ndarray:
    """librosa.feature.mfcc(S=log_mel, n_mfcc=n): orthonormal DCT-II over the mel axis, first n rows."""
    return scipy.fftpack.dct(log_mel, axis=-2, type=2, norm="ortho")[:n_mfcc, :]


def normalize_statics(c: np.ndarray) -> np.ndarray:
    """mfcc.py:50-69: axis=0 on a (13, T) array, i.e. per FRAME across coefficients."""
    mean = np.mean(c, axis=0, keepdims=True)
    std = np.std(c, axis=0, keepdims=True)
    return (c - mean) / (std + 1e-8)


def mfcc_features(y: np.ndarray, sr: float = 16000, n_mfcc: int = N_MFCC) -> np.ndarray:
    """mfcc.py:24-44: float32 [3*n_mfcc, T]."""
    if not isinstance(y, np.ndarray):
        raise TypeError("Input signal must be a numpy array.")
    if y.ndim != 1:
        raise ValueError("Input signal must be 1-dimensional.")
    log_mel = power_to_db(mel_power(y, sr))
    c = dct_mfcc(log_mel, n_mfcc)
    d1 = delta(c, 1)
    d2 = delta(c, 2)
    return np.concatenate((normalize_statics(c), d1, d2), axis=0)


def mfcc_batch(signals: List[np.ndarray], sr: float = 16000) -> List[np.ndarray]:
    """mfcc.py:71-84: transposed (T, 39) per utterance."""
    return [mfcc_features(s, sr).T for s in signals]
