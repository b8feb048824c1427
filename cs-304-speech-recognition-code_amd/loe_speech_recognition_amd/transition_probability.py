"""Sparse transition stores with the reference's semantics
(src/loe_speech_recognition/transition_probability.py:11-82): a dict keyed by (from, to);
a key that was never stored reads as 0.0; log-matrices stack block-diagonally.

These are small host-side containers.  The device works on the dense 3-band form produced
by ``band`` below."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Optional, Sequence, Tuple

import numpy as np


_KEYS: Dict[int, list] = {}


def _dense_keys(n: int) -> list:
    keys = _KEYS.get(n)
    if keys is None:
        keys = _KEYS[n] = [(i, j) for i in range(n) for j in range(n)]
    return keys


@dataclass
class SparseMatrix:
    num_of_states: int = 0
    _core: Dict[Tuple[int, ...], float] = field(default_factory=dict)

    def __getitem__(self, key: Tuple[int, ...]) -> float:
        return self._core.get(tuple(key), 0.)

    def __setitem__(self, key: Tuple[int, ...], value: float) -> None:
        self._core[tuple(key)] = value
        self.__dict__.pop("_band_cache", None)

    def _fill_dense(self, dense: np.ndarray, keep: Optional[np.ndarray] = None) -> None:
        """Store every entry of a dense [N, N] array (those where ``keep`` is True) under its (row, column) key,
        row by row like a double loop would, without the per-entry Python work."""
        n = dense.shape[0]
        keys = _dense_keys(n)
        values = list(dense.reshape(-1))                       # NumPy scalars of the array's dtype, as indexing yields
        if keep is None:
            self._core.update(zip(keys, values))
        else:
            self._core.update((k, v) for k, v, ok in zip(keys, values, keep.reshape(-1).tolist()) if ok)
        self.__dict__.pop("_band_cache", None)

    def __str__(self) -> str:
        return f"{self._core}"

    def to_array(self) -> np.ndarray:
        """Dense float32 [N, N]; entries that were never stored are NaN."""
        out = np.full((self.num_of_states, self.num_of_states), np.nan, dtype=np.float32)
        for (i, j), v in self._core.items():
            out[i, j] = v
        return out


@dataclass
class TransitionProbabilities(SparseMatrix):

    @classmethod
    def from_num_of_states(cls, num_of_states: int) -> "TransitionProbabilities":
        """Row i spreads probability 1/(N-i) over states i..N-1 (transition_probability.py:42-52)."""
        dense = np.zeros((num_of_states, num_of_states), dtype=np.float32)
        for i in range(num_of_states):
            dense[i, i:] = 1 / (num_of_states - i)
        return cls.from_transition_probability(dense)

    @classmethod
    def from_transition_probability(cls, transition_probability: np.ndarray) -> "TransitionProbabilities":
        """Every entry of the dense matrix is stored, zeros included (transition_probability.py:54-64)."""
        tp = cls(transition_probability.shape[0])
        dense = np.asarray(transition_probability)
        tp._fill_dense(dense, keep=dense != -float("inf"))
        return tp


@dataclass
class LogTransitionProbabilities(SparseMatrix):

    def append(self, ltp: "LogTransitionProbabilities") -> None:
        base = self.num_of_states
        self.num_of_states += ltp.num_of_states
        for point, value in ltp._core.items():
            self._core[tuple(i + base for i in point)] = value
        self.__dict__.pop("_band_cache", None)

    @classmethod
    def from_transition_probability(cls, tp: TransitionProbabilities) -> "LogTransitionProbabilities":
        ltp = cls(tp.num_of_states)
        if not tp._core:
            return ltp
        # one vectorised logarithm over the stored values (same dtype, same bits as value by value)
        values = np.array(list(tp._core.values()))
        with np.errstate(divide="ignore", invalid="ignore"):
            logs = np.log(values)
        ltp._core.update(zip(tp._core.keys(), list(logs)))
        return ltp

    def band(self, lower_of_state: Optional[Sequence[int]] = None) -> np.ndarray:
        """float32 [3, N] with band[d, s] = log a(s-d -> s): the only entries the left-to-right
        recursions read (hidden_markov_model.py:181-184, 518-521).  Keys that were never stored
        read 0.0 like ``__getitem__``; predecessors below 0, or below the word's first state when
        ``lower_of_state`` is given, are -inf."""
        n = self.num_of_states
        key = None if lower_of_state is None else tuple(int(v) for v in lower_of_state)
        cached = self.__dict__.get("_band_cache")
        if cached is not None and cached[0] == key and cached[1] == len(self._core):
            return cached[2].copy()
        out = np.full((3, n), -np.inf, dtype=np.float32)
        for s in range(n):
            floor = 0 if lower_of_state is None else int(lower_of_state[s])
            for d in range(3):
                o = s - d
                if o >= floor and o >= 0:
                    out[d, s] = np.float32(self[(o, s)])
        self.__dict__["_band_cache"] = (key, len(self._core), out.copy())
        return out
