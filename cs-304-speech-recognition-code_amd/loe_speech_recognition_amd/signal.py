"""Aligned utterances and their grouping by state (src/loe_speech_recognition/signal.py).
Kept for API compatibility: the trainers here reduce alignments on the device from per-frame
state slots and do not build these objects on their hot path."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

from .transition_probability import TransitionProbabilities


@dataclass
class Signal:
    num_of_state: int
    signal: np.ndarray          # float32 [T, 39]
    path: np.ndarray            # int8 [T]

    @property
    def order_by_state(self) -> List[Optional[np.ndarray]]:
        """signal.py:23-47: states are visited in ascending order; each takes the run of frames
        that carry it, starting where the previous state stopped (None when it has none)."""
        out: List[Optional[np.ndarray]] = []
        start = 0
        for state in range(self.num_of_state):
            end = start
            while end < len(self.path) and self.path[end] == state:
                end += 1
            out.append(self.signal[start:end] if end > start else None)
            start = end
        return out

    @property
    def order_by_signal(self) -> List[Tuple[np.ndarray, int]]:
        return list(zip(self.signal, self.path))


@dataclass
class SortedSignals:
    num_of_states: int
    _signals: List[Signal] = field(default_factory=list)

    def append(self, signal: Signal) -> None:
        self._signals.append(signal)

    @property
    def order_by_state(self) -> List[List[np.ndarray]]:
        by_state: List[List[np.ndarray]] = [[] for _ in range(self.num_of_states)]
        for sig in self._signals:
            for state, part in enumerate(sig.order_by_state):
                if part is not None:
                    by_state[state].append(part)
        return by_state

    @property
    def transition_counts(self) -> np.ndarray:
        counts = np.zeros((self.num_of_states, self.num_of_states), dtype=np.int32)
        for sig in self._signals:
            p = np.asarray(sig.path, dtype=np.int64)
            np.add.at(counts, (p[:-1], p[1:]), 1)
        return counts

    @property
    def transition_probabilities(self) -> TransitionProbabilities:
        """signal.py:80-91: row-normalised transition counts (rows without counts become NaN)."""
        counts = self.transition_counts
        with np.errstate(invalid="ignore", divide="ignore"):
            probs = (counts / np.sum(counts, axis=1, keepdims=True)).astype(np.float32)
        return TransitionProbabilities.from_transition_probability(probs)
