"""CPU restatement of the reference's energy-based silence stripper — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product path (liblsr_hip + loe_speech_recognition_amd) never does.

PINNED: tests/golden/sep_g1.npz holds outputs of the reference's own SignalSeparation
(src/loe_speech_recognition/signal_separation.py), produced in the build container by
tools/make_golden_separation.py; tests/test_oracle_separation.py requires this restatement to
reproduce them exactly (slice bounds, failures, noise clips including the carry-over between calls).

What the reference does (signal_separation.py:88-164), restated:

* frames of ``frame_size = int(sample_rate * frame_time)`` samples, plus one trailing frame that
  holds the remaining samples and may be EMPTY (its mean is NaN and compares False, so it counts
  as a frame without speech) (:103-109);
* frame energy = ``np.average(np.abs(frame))``: a float32 mean (NumPy's pairwise float32 sum
  divided by the count in float32), compared against ``threshold * max|signal|`` — an np.float64
  (the maximum is ``np.max(...).astype(float)``), so the comparison runs in float64 (:71-76,105,153-164);
* a two-threshold state machine: speech starts above the high threshold, continues above the
  low one; frames from the first start on are kept; ``maximum_silence_frames`` consecutive
  frames without speech end the clip, and the frame that trips the counter is not kept (:117-145);
* frames seen outside speech below the high threshold are noise; the noise list is only cleared
  when a clip ends normally, so noise of clips that failed is carried into the next one (:92-93,139);
* a clip that never trips the counter fails, and so does one with fewer than 9 kept frames —
  after its noise has been filed (:94-100).
"""
from __future__ import annotations

import warnings
from typing import List, Optional, Tuple

import numpy as np

OK, NO_END, TOO_SHORT = 0, 1, 2
MIN_KEPT_FRAMES = 9          # signal_separation.py:95


def frame_energies(signal: np.ndarray, frame_size: int) -> np.ndarray:
    """float32 mean |x| of every scan frame, the trailing (possibly empty -> NaN) one included."""
    n_full = signal.shape[0] // frame_size
    out = np.empty(n_full + 1, np.float32)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for k in range(n_full):
            out[k] = np.average(np.abs(signal[k * frame_size:(k + 1) * frame_size]))
        out[n_full] = np.average(np.abs(signal[n_full * frame_size:]))
    return out


def scan(energy: np.ndarray, max_volume: float, high: float, low: float,
         max_silence_frames: int) -> Tuple[int, int, int, np.ndarray]:
    """(first kept frame, end frame (exclusive), status, noise flag per scan frame).

    Statuses: OK, NO_END (the counter never tripped: signal_separation.py:100), TOO_SHORT (:95-97).
    first/end are -1 where they do not exist."""
    # _max_volume is np.max(...).astype(float), an np.float64, so threshold * _max_volume is an np.float64 too
    # and `average_energy > threshold` promotes the float32 energy to float64 (signal_separation.py:71-76,159-161)
    thr_high = np.float64(high) * np.float64(max_volume)
    thr_low = np.float64(low) * np.float64(max_volume)
    n = energy.shape[0]
    noise = np.zeros(n, np.uint8)
    between = ever = False
    counter = 0
    first = end = -1
    for k in range(n):
        e = energy[k]
        tripped = False
        if between:
            if e > thr_low:
                counter = 0
            else:
                between = False
                counter += 1
                tripped = counter >= max_silence_frames
        else:
            if e > thr_high:
                between = True
                if not ever:
                    first = k
                ever = True
                counter = 0
            else:
                noise[k] = 1
                if ever:
                    counter += 1
                    tripped = counter >= max_silence_frames
        if tripped:
            end = k
            break
    if end < 0:
        return first, -1, NO_END, noise
    return first, end, (TOO_SHORT if end - first < MIN_KEPT_FRAMES else OK), noise


class SignalSeparationRef:
    """Stateful like the reference object: noise of failed clips is carried over (:57-60,92-93)."""

    def __init__(self, sample_rate: int = 16000, frame_time: float = 0.01, speech_high_threshold: float = 0.08,
                 speech_low_threshold: float = 0.01, silence_duration_threshold: float = 0.02):
        self.frame_size = int(sample_rate * frame_time)
        self.maximum_silence_frames = int(silence_duration_threshold / frame_time)
        self.high = speech_high_threshold
        self.low = speech_low_threshold
        self.noises: List[np.ndarray] = []
        self._carry: List[np.ndarray] = []

    def remove_empty(self, signal: np.ndarray) -> Tuple[int, Optional[np.ndarray]]:
        """(status, kept samples or None)."""
        fs = self.frame_size
        energy = frame_energies(signal, fs)
        max_volume = float(np.max(np.abs(signal)))
        first, end, status, noise = scan(energy, max_volume, self.high, self.low, self.maximum_silence_frames)
        for k in np.nonzero(noise)[0]:
            self._carry.append(signal[k * fs:(k + 1) * fs])
        if status == NO_END:
            return status, None
        self.noises.append(np.concatenate(self._carry, dtype=np.float32))
        self._carry = []
        if status == TOO_SHORT:
            return status, None
        return status, signal[first * fs:end * fs].astype(np.float32)

    def remove_empty_batch(self, signals: List[np.ndarray]) -> List[np.ndarray]:
        out = []
        for s in signals:
            status, kept = self.remove_empty(s)
            if status == OK:
                out.append(kept)
        return out
