"""MFCC front-end with the reference's interface (src/loe_speech_recognition/mfcc.py:12-84),
computed on the GPU by liblsr_hip (lsr_mfcc_batch)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List

import numpy as np

from . import _native


def _validate(signal, sample_rate, n_mfcc) -> None:
    if not isinstance(signal, np.ndarray):
        raise TypeError("Input signal must be a numpy array.")
    if signal.ndim != 1:
        raise ValueError("Input signal must be 1-dimensional.")
    if not np.issubdtype(signal.dtype, np.floating):
        raise ValueError("Audio data must be floating-point")        # librosa.util.valid_audio
    if float(sample_rate) != 16000.0:
        raise ValueError("the device front-end is built for 16 kHz audio (mel filterbank of mfcc.py:31-34)")
    if n_mfcc != 13:
        raise ValueError("the device front-end produces 13 cepstra (+13 deltas +13 delta-deltas)")


@dataclass
class MFCC:
    # Input
    signal: np.ndarray
    sample_rate: int | float

    # Settings
    n_mfcc: int = field(default=13)

    # Internals
    _feature_vector: np.ndarray = field(init=False)

    def __post_init__(self) -> None:
        _validate(self.signal, self.sample_rate, self.n_mfcc)
        self._feature_vector = self.batch([self.signal], self.sample_rate)[0].T

    @property
    def feature_vector(self) -> np.ndarray:
        """float32 (39, T): normalised statics, deltas, delta-deltas (mfcc.py:43)."""
        return self._feature_vector

    @staticmethod
    def normalize_mfccs(mfccs):
        """mfcc.py:50-69 for a caller-supplied array (the batch kernels do this on the device)."""
        mean = np.mean(mfccs, axis=0, keepdims=True)
        std = np.std(mfccs, axis=0, keepdims=True)
        return (mfccs - mean) / (std + 1e-8)

    @classmethod
    def batch(cls, signals: List[np.ndarray], sample_rate: int) -> List[np.ndarray]:
        """**Transposed** feature matrices, one float32 (T, 39) array per clip (mfcc.py:71-84),
        from a single device call over the whole list."""
        if len(signals) == 0:
            return []
        for s in signals:
            _validate(s, sample_rate, 13)
        eng = _native.get_engine()
        ab = eng.audio_batch(signals)
        fb = eng.mfcc(ab)
        return _native.split_ragged(eng.to_host(fb.feats), fb.host_off)

    @classmethod
    def batch_int16(cls, signals: List[np.ndarray], sample_rate: int = 16000) -> List[np.ndarray]:
        """``batch`` for int16 PCM straight from ``scipy.io.wavfile.read``: the float32 conversion the
        reference's loader does (ti_digits.py:130-134) happens on the device."""
        if len(signals) == 0:
            return []
        for s in signals:
            if not isinstance(s, np.ndarray) or s.ndim != 1 or s.dtype != np.int16:
                raise TypeError("batch_int16 expects 1-dimensional int16 arrays")
        if float(sample_rate) != 16000.0:
            raise ValueError("the device front-end is built for 16 kHz audio")
        eng = _native.get_engine()
        fb = eng.mfcc(eng.audio_batch(signals))
        return _native.split_ragged(eng.to_host(fb.feats), fb.host_off)

    @classmethod
    def batch_device(cls, signals: List[np.ndarray], sample_rate: int = 16000) -> "_native.FeatureBatch":
        """Same as ``batch`` but the features stay on the device for the decoders/trainers."""
        for s in signals:
            _validate(s, sample_rate, 13)
        eng = _native.get_engine()
        return eng.mfcc(eng.audio_batch(signals))
