"""Energy-based silence stripping with the reference's interface
(src/loe_speech_recognition/signal_separation.py:42-164); the per-frame energies and the
two-threshold scan of every clip of a batch run on the GPU (lsr_separation_scan)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Literal, Tuple

import numpy as np

from . import _native

KEPT, NEVER_ENDED, TOO_SHORT = 0, 1, 2
MIN_KEPT_FRAMES = 9            # "This threshold is based on MFCC" (signal_separation.py:95)


@dataclass
class SignalSeparation:
    class FailToProcess(Exception):
        pass

    # Settings
    sample_rate: int = field(default=16000)
    frame_time: float = field(default=0.01)
    speech_high_threshold: float = field(default=0.08)
    speech_low_threshold: float = field(default=0.01)
    silence_duration_threshold: float = field(default=0.02)

    # Internals
    _noises: List[np.ndarray] = field(default_factory=list)
    _max_volume: float = field(init=False, default=0.0)
    _noise: List[np.ndarray] = field(default_factory=list)      # noise of clips that failed, not yet filed

    @property
    def frame_size(self) -> int:
        return int(self.sample_rate * self.frame_time)

    @property
    def maximum_silence_frames(self) -> int:
        return int(self.silence_duration_threshold / self.frame_time)

    # -- the device scan ---------------------------------------------------------------------------
    def _scan(self, signals: List[np.ndarray]) -> dict:
        for s in signals:
            if not isinstance(s, np.ndarray) or s.ndim != 1:
                raise TypeError("signals must be 1-dimensional numpy arrays")
        eng = _native.get_engine()
        return eng.separation_scan(signals, self.frame_size, self.speech_high_threshold, self.speech_low_threshold,
                                   self.maximum_silence_frames, MIN_KEPT_FRAMES)

    def _file_noise(self, signal: np.ndarray, flags: np.ndarray, status: int) -> None:
        """Noise frames join the pending list; a clip that ended normally files the list as one clip
        (signal_separation.py:92-93,139) — so noise of failed clips is carried into the next one."""
        fs = self.frame_size
        idx = np.nonzero(flags)[0]
        if idx.shape[0]:
            # runs of consecutive noise frames are contiguous slices of the clip
            breaks = np.nonzero(np.diff(idx) != 1)[0] + 1
            for run in np.split(idx, breaks):
                self._noise.append(signal[run[0] * fs:(run[-1] + 1) * fs])
        if status != NEVER_ENDED:
            self._noises.append(np.concatenate(self._noise, dtype=np.float32))   # ValueError when empty, as the reference
            self._noise = []

    def _results(self, signals: List[np.ndarray], scan: dict) -> List[Tuple[int, int, int]]:
        fs = self.frame_size
        out = []
        for i, s in enumerate(signals):
            lo, hi = scan["scan_off"][i], scan["scan_off"][i + 1]
            status = int(scan["status"][i])
            self._max_volume = float(scan["max_volume"][i])
            self._file_noise(s, scan["noise"][lo:hi], status)
            out.append((status, int(scan["first"][i]) * fs, int(scan["end"][i]) * fs))
        return out

    # -- reference interface -----------------------------------------------------------------------
    def remove_empty_batch(self, signals: List[np.ndarray]) -> List[np.ndarray]:
        """Kept part of every clip that could be processed, failures dropped (signal_separation.py:78-86);
        one device call for the whole list."""
        if len(signals) == 0:
            return []
        scan = self._scan(signals)
        return [np.asarray(s[a:b], dtype=np.float32) for s, (status, a, b) in zip(signals, self._results(signals, scan))
                if status == KEPT]

    def remove_empty(self, signal: np.ndarray) -> np.ndarray:
        scan = self._scan([signal])
        status, a, b = self._results([signal], scan)[0]
        if status != KEPT:
            raise self.FailToProcess
        return np.asarray(signal[a:b], dtype=np.float32)

    def remove_empty_batch_device(self, signals: List[np.ndarray]) -> Tuple["_native.AudioBatch", List[int]]:
        """(kept clips as a device batch ready for the MFCC kernels, indices of the clips kept): the
        kept samples are gathered on the device and never cross the bus again."""
        scan = self._scan(signals)
        res = self._results(signals, scan)
        keep = [i for i, (status, _, _) in enumerate(res) if status == KEPT]
        if not keep:
            raise self.FailToProcess("no clip could be processed")
        eng = _native.get_engine()
        start = np.array([scan["sample_off"][i] + res[i][1] for i in keep], dtype=np.int64)
        length = np.array([res[i][2] - res[i][1] for i in keep], dtype=np.int64)
        audio, soff = eng.gather_segments(scan["audio"], start, length)
        foff = np.concatenate([[0], np.cumsum(1 + length // _native.HOP)]).astype(np.int64)
        return _native.AudioBatch(audio, eng.to_device(soff), eng.to_device(foff), length), keep

    def get_all_noises(self) -> List[np.ndarray]:
        return self._noises

    def detect_speech(self, frames: np.ndarray, threshold: Literal["high", "low"]) -> bool:
        """Host form of the per-frame test (signal_separation.py:153-164) against the maximum of the
        clip scanned last; the batch methods evaluate it on the device."""
        average_energy = np.average(np.abs(frames))
        if threshold == "high" and average_energy > self.speech_high_threshold * self._max_volume:
            return True
        elif threshold == "low" and average_energy > self.speech_low_threshold * self._max_volume:
            return True
        return False
