"""Word boundaries of a stacked state space and the path -> label rule
(src/loe_speech_recognition/model_boundary.py).  Host-side bookkeeping; the batched decoder
applies the same rule on the device and this class is its per-utterance counterpart."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Tuple

import numpy as np


@dataclass
class ModelBoundary:
    _boundaries: List[int] = field(default_factory=list)     # cumulative state counts
    _labels: List[str] = field(default_factory=list)
    _isFrozen: bool = False

    @property
    def lower_boundaries(self) -> List[int]:
        self._isFrozen = True
        return sorted([0] + self._boundaries[:-1])

    @property
    def upper_boundaries(self) -> List[int]:
        self._isFrozen = True
        return sorted(b - 1 for b in self._boundaries)

    @property
    def num_of_words(self) -> int:
        return len(self._boundaries)

    @property
    def num_of_states(self) -> int:
        return self._boundaries[-1] if self._boundaries else 0

    def append(self, num_of_states: int) -> None:
        if self._isFrozen:
            raise Exception("Adding data after cache is created")
        self._boundaries.append((self._boundaries[-1] if self._boundaries else 0) + num_of_states)

    def find_lower_boundary(self, state: int) -> int:
        for lb in reversed(self.lower_boundaries):
            if state >= lb:
                return lb
        raise Exception(f"Failed to find lower boundary for state {state}")

    def find_upper_boundary(self, state: int) -> int:
        for ub in self.upper_boundaries:
            if state <= ub:
                return ub
        raise Exception(f"Failed to find upper boundary for state {state}")

    def add_model_labels(self, model_labels: List[str]) -> None:
        assert len(model_labels) == self.num_of_words
        self._labels = model_labels

    def get_label(self, state: int) -> str:
        return self._labels[self.lower_boundaries.index(self.find_lower_boundary(state))]

    def get_state_range(self, label: str) -> Tuple[int, int]:
        i = self._labels.index(label)
        return (0 if i == 0 else self._boundaries[i - 1], self._boundaries[i])

    def append_to_labels(self, state: int, skip_silence: bool, labels: List[str]) -> None:
        label = self.get_label(state)
        if not (label == "S" and skip_silence):
            labels.append(label)

    def get_labels(self, path: np.ndarray, skip_silence: bool = True) -> List[str]:
        """model_boundary.py:107-139.  A word is reported for the first run of the path, whenever
        the state leaves the current word's range, and when the path jumps from a word's last
        state back to the same word's first state (a repeated word)."""
        states = np.asarray(path).tolist()
        runs = [states[0]]
        for s in states[1:]:
            if s != runs[-1]:
                runs.append(s)
        labels: List[str] = []
        lo = self.find_lower_boundary(runs[0])
        hi = self.find_upper_boundary(runs[0])
        self.append_to_labels(runs[0], skip_silence, labels)
        for i in range(1, len(runs)):
            cur = runs[i]
            if cur < lo or cur > hi:
                lo = self.find_lower_boundary(cur)
                hi = self.find_upper_boundary(cur)
                self.append_to_labels(cur, skip_silence, labels)
            elif runs[i - 1] == hi and cur == lo:
                self.append_to_labels(cur, skip_silence, labels)
        return labels
